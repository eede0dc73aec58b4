"""var_amd: MI355X-native next-scale sampling path of VAR."""
__version__ = '0.1.0'
