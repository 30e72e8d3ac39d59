from .flat import FlatParams, FusedAdam

__all__ = ["FlatParams", "FusedAdam"]
