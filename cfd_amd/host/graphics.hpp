// graphics.hpp - the globals a generated main_setup() reads (FluidX3D-master/src/graphics.cpp:3-19, graphics.hpp:13-14,
// 423-426).  There is no window: `running` stays true until the setup returns, `key_P` (run/pause) starts true, and
// `key_9` (export trigger) is raised by whoever started the process: a character on standard input or a signal (main.cpp),
// or directly by code linked into the same process (fluid_wrapper, a test).
#pragma once
#include <atomic>
#include "utilities.hpp"

extern std::atomic_bool running;
extern std::atomic_bool key_P;
extern std::atomic_bool key_9;
extern vector<string> main_arguments; // device ordinals from the command line (DOCUMENTATION.md:132)
