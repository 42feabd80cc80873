__all__ = ["homo_constr"]
