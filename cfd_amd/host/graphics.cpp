// graphics.cpp - nothing to compile: there is no renderer; the key and run flags live in lbm.cpp and main.cpp.
// The file exists so that the source list of the GUI's build line (main.py:307-319) can be used as it is.
#include "graphics.hpp"
