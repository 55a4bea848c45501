"""Alias package: the reference's `unet.*` dotted paths -> adm_amd.unet.* (SURVEY.md section 8b)."""
