"""ctypes binding of libcgamd.so (the C-ABI in include/cgamd.h) + torch plumbing around it."""
