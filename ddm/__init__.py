"""Alias package: the reference's `ddm.*` dotted paths -> adm_amd.ddm.* (SURVEY.md section 8b)."""
