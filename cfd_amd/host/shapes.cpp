// shapes.cpp - nothing to compile: the geometry predicates are inline (shapes.hpp).  Present for the GUI's source list.
#include "shapes.hpp"
