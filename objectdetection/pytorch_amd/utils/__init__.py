"""Drop-in for the reference's top-level `utils` package (hot-path helpers only: utils/utils.py)."""
