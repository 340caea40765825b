"""CPU oracle of the torchfx.filter hot path -- TEST INFRASTRUCTURE ONLY (never imported by torchfx_amd)."""
