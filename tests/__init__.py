"""Test suite of torchfx_amd: CPU tests (-m "not gpu") and MI355X parity tests (-m gpu)."""
