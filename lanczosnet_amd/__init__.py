"""MI355X-native LanczosNet hot path.  Importing the package is cheap and works
without a GPU; the HIP library is loaded on first use and its absence is a hard error."""
__version__ = '0.1.0'
