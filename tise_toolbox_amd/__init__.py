"""MI355X-native image-realism hot path of the TISE toolbox (IS* / FID)."""
__version__ = "0.1.0"
