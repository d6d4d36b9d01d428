__all__ = ["bias", "dispersion", "predict"]
