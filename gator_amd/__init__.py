"""gator_amd: MI355X-native (gfx950) implementation of the GATOR forward path behind the reference's lib/models API."""
__version__ = '0.1.0'
