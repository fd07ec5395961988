"""Drop-in for the native module `simple_knn._C` (KNN/ext.cpp:15-17)."""
from s3gaussian_amd.knn import distCUDA2  # noqa: F401
