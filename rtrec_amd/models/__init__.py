from .slim import SLIM

__all__ = ["SLIM"]
