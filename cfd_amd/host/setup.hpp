// setup.hpp - what a setup.cpp includes (FluidX3D-master/src/setup.hpp:1-7)
#pragma once
#include "defines.hpp"
#include "lbm.hpp"
void main_setup();
