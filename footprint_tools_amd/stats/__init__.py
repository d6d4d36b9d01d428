__all__ = ["distributions", "windowing", "posterior", "utils", "fdr"]
