"""Mirror of the reference's `architectures` package for the hot path."""
