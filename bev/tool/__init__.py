__all__ = ["compo"]
