// info.cpp - nothing to compile: Info is header-only here (info.hpp).  Present for the GUI's source list (main.py:307-319).
#include "info.hpp"
