"""Drop-in for the reference's `simple_knn` package (KNN/setup.py:22-25), backed by libs3g.so."""
