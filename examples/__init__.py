"""Simulation scripts written against the `sailfish` API, run through the HIP backend."""
