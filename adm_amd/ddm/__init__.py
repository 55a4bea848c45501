"""HIP-backed DDM diffusion wrappers with the reference's module paths (ddm.ddm_const*, DDPM)."""
