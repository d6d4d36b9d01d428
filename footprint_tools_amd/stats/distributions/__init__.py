__all__ = ["nbinom"]
