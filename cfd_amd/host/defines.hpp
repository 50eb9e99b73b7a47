// defines.hpp - compile-time configuration of the C++ host, same switch names as the reference's defines.hpp
// (FluidX3D-master/src/defines.hpp.bak:5-105).  Select one velocity set, one collision operator, optionally one
// 16-bit DDF format, and any of the extensions; pass them with -D on the compiler command line or edit here.
#pragma once

#if !defined(D2Q9) && !defined(D3Q15) && !defined(D3Q19) && !defined(D3Q27)
#define D3Q19
#endif
#if !defined(SRT) && !defined(TRT)
#define SRT
#endif
// optional: FP16S, FP16C, UPDATE_FIELDS, VOLUME_FORCE, MOVING_BOUNDARIES, EQUILIBRIUM_BOUNDARIES, FORCE_FIELD, SUBGRID
// BENCHMARK switches every extension off, as in the reference (defines.hpp.bak)
#ifdef BENCHMARK
#undef UPDATE_FIELDS
#undef VOLUME_FORCE
#undef MOVING_BOUNDARIES
#undef EQUILIBRIUM_BOUNDARIES
#undef FORCE_FIELD
#undef SUBGRID
#endif
// the reference couples interactive graphics to UPDATE_FIELDS (defines.hpp:49-51); there is no window here, but a setup
// that asks for it keeps its field-update timing
#if defined(INTERACTIVE_GRAPHICS) || defined(INTERACTIVE_GRAPHICS_ASCII)
#define UPDATE_FIELDS
#endif

// cell flags (FluidX3D-master/src/defines.hpp:25-32)
#define TYPE_S 0x01 // solid boundary (stationary or moving)
#define TYPE_E 0x02 // equilibrium boundary (inflow/outflow)
#define TYPE_T 0x04 // temperature boundary (unused here)
#define TYPE_F 0x08 // fluid (free-surface extension, unused here)
#define TYPE_I 0x10 // interface (unused here)
#define TYPE_G 0x20 // gas (unused here)
#define TYPE_X 0x40 // free for the setup
#define TYPE_Y 0x80 // free for the setup

// visualisation mode bits are accepted and ignored (there is no renderer)
#define VIS_FLAG_LATTICE 0x01
#define VIS_FLAG_SURFACE 0x02
#define VIS_FIELD 0x04
#define VIS_STREAMLINES 0x08
#define VIS_Q_CRITERION 0x10
#define VIS_PHI_RASTERIZE 0x20
#define VIS_PHI_RAYTRACE 0x40
#define VIS_PARTICLES 0x80

#if defined(FP16S) || defined(FP16C)
#define fpxx ushort
#else
#define fpxx float
#endif
