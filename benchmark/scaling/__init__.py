"""Weak-scaling harness (one block per GPU)."""
