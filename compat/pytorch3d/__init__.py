"""Import-namespace stand-in for pytorch3d (see compat/README.md).  Not pytorch3d: only the names yifita/DSS imports,
re-implemented in plain PyTorch from pytorch3d's published API and conventions."""
__version__ = "0.4.0+dss_amd.compat"
