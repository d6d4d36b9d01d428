"""footprint_tools_amd -- MI355X (gfx950) implementation of the footprint-tools
per-nucleotide expected-cleavage / deviation-statistics scan.

Drop-in for the `footprint_tools.modeling` / `footprint_tools.stats` call surface of
vierstralab/footprint-tools v1.3.7 on that path::

    from footprint_tools_amd.modeling import bias, predict, dispersion
    from footprint_tools_amd.stats import windowing, fdr, utils, posterior

plus the batched, HBM-resident scan in :mod:`footprint_tools_amd.scan`.  All arithmetic
runs in hand-written HIP kernels behind the C ABI of ``include/fpt.h``
(``libfpt_hip.so``); there is no CPU fallback.
"""
__version__ = "0.1.0"
__all__ = ["modeling", "stats", "scan", "detect", "distributed"]
