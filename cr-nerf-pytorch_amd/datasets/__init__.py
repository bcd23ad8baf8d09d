"""Mirror of the part of the reference's ``datasets`` package that feeds the hot path: ray generation
(datasets/ray_utils.py).  Dataset readers / COLMAP I/O are out of scope."""
