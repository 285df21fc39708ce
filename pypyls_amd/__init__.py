"""pypyls_amd -- MI355X-native PLS-C resampling engine behind the pyls front-ends."""
__version__ = '0.1.0'
