"""Test suite: oracle / host-layer tests (CPU) and HIP parity tests (-m gpu)."""
