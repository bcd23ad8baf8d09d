"""Host-side mirror of the reference's ``models`` package for the rendering hot path: same module
paths (models.rendering / models.nerf / models.linearStyleTransfer / models.nerf_decoder_stylenerf),
same class/function names, constructor signatures and state_dict keys -- HIP kernels underneath."""
