"""sailfish_amd -- MI355X-native backend + host layer for the Sailfish LB hot path."""
__version__ = '0.1.0'
