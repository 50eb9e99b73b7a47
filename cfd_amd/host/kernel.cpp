// kernel.cpp - nothing to compile: the kernels are HIP code inside liblbmhip.so (cfd_amd/csrc), not OpenCL text built at
// run time.  Present for the GUI's source list (main.py:307-319).
