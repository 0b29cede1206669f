"""Simulation configuration (reference sailfish/config.py)."""
import argparse
import configparser
import os


class LBConfig(argparse.Namespace):
    """Option namespace; dynamic attributes as in the reference (config.py:17-30)."""

    @property
    def output_required(self):
        return bool(getattr(self, 'output', '')) or getattr(self, 'mode', 'batch') == 'visualization'

    @property
    def needs_iteration_num(self):
        return bool(getattr(self, 'time_dependence', False)) or getattr(self, 'access_pattern', 'AB') == 'AA'


class LBConfigParser(object):
    """argparse front-end with option groups and rc files (reference config.py:33-91:
    /etc/sailfishrc, ~/.sailfishrc, .sailfishrc)."""

    def __init__(self, description=None):
        self._parser = argparse.ArgumentParser(description=description)
        self._parser.add_argument('-q', '--quiet', help='reduce verbosity', action='store_true', default=False)
        self._parser.add_argument('-v', '--verbose', help='print additional info about the simulation',
                                  action='store_true', default=False)
        self._parser.add_argument('--silent', action='store_true', default=False,
                                  help='no output to stdout')
        self.config = LBConfig()

    def add_group(self, name):
        return self._parser.add_argument_group(name)

    def set_defaults(self, defaults):
        """Values for declared options, and attributes for names no option declares (simulation scripts pass
        private settings this way; the reference asserts that every key is a declared option, config.py:51-54)."""
        return self._parser.set_defaults(**defaults)

    def parse(self, args, internal_defaults=None):
        rc = configparser.ConfigParser()
        rc.read(['/etc/sailfishrc', os.path.expanduser('~/.sailfishrc'), '.sailfishrc'])
        try:
            self._parser.set_defaults(**dict(rc.items('main')))
        except configparser.NoSectionError:
            pass
        if internal_defaults is not None:
            self._parser.set_defaults(**internal_defaults)
        self._parser.parse_args(args=args, namespace=self.config)
        # Additional internal config options, not settable via the command line.
        self.config.relaxation_enabled = getattr(self.config, 'relaxation_enabled', True)
        self.config.propagation_enabled = getattr(self.config, 'propagation_enabled', True)
        self.config.time_dependence = False
        self.config.space_dependence = False
        self.config.unit_test = False
        return self.config
