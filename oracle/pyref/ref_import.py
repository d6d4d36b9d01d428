"""DEV-CONTAINER TOOLING (test infrastructure): import the genuine reference
package from /root/reference with the extensions built by build_pyref.py.

The reference's top-level __init__ needs installed dist metadata and its
modules import third-party packages at module scope that are not installed here
and not used on the hot path (genome_tools, pwlf, simplejson); those are
replaced by minimal stand-ins so that the reference's OWN code for this path
runs unmodified.
"""
import json
import os
import sys
import types

REF = os.environ.get("FPT_REFERENCE", "/root/reference")
OUT = os.environ.get("FPT_PYREF_OUT", "/tmp/fpt_pyref")


class genomic_interval(object):
    """Duck-type of genome_tools.genomic_interval (chrom/start/end/widen/len)."""

    def __init__(self, chrom, start, end, name=".", score=None, strand=None):
        self.chrom, self.start, self.end = str(chrom), int(start), int(end)
        self.name, self.score, self.strand = name, score, strand

    def __len__(self):
        return self.end - self.start

    def widen(self, w):
        return genomic_interval(self.chrom, self.start - w, self.end + w, self.name,
                                self.score, self.strand)


def load():
    if "footprint_tools" in sys.modules and getattr(sys.modules["footprint_tools"], "_fpt_pyref", 0):
        return sys.modules["footprint_tools"]
    pkg = types.ModuleType("footprint_tools")
    pkg.__path__ = [os.path.join(REF, "footprint_tools"), os.path.join(OUT, "pkg", "footprint_tools")]
    pkg.__version__ = "1.3.7"
    pkg._fpt_pyref = 1
    sys.modules["footprint_tools"] = pkg
    for sub in ("modeling", "stats", "stats/distributions", "stats/fdr"):
        name = "footprint_tools." + sub.replace("/", ".")
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, "footprint_tools", sub),
                      os.path.join(OUT, "pkg", "footprint_tools", sub)]
        sys.modules[name] = m
    gt = types.ModuleType("genome_tools")
    gt.genomic_interval = genomic_interval
    sys.modules.setdefault("genome_tools", gt)
    sys.modules.setdefault("pwlf", types.ModuleType("pwlf"))
    sys.modules.setdefault("simplejson", json)
    # stats/fdr is a package whose __init__ holds the code: execute it for real
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "footprint_tools.stats.fdr", os.path.join(REF, "footprint_tools/stats/fdr/__init__.py"),
        submodule_search_locations=[os.path.join(REF, "footprint_tools/stats/fdr")])
    fdr = importlib.util.module_from_spec(spec)
    sys.modules["footprint_tools.stats.fdr"] = fdr
    spec.loader.exec_module(fdr)
    return pkg
