// lodepng.cpp - nothing to compile: no images are written (no renderer).  Present for the GUI's source list (main.py:307-319).
