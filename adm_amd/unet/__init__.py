"""HIP-backed UNets with the reference's module paths (unet.uncond_unet*, EDMPrecond)."""
