"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's algorithm for the rendering hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (cr-nerf-pytorch_amd/) never does."""
