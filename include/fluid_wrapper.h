/*
 * fluid_wrapper.h - the four symbols the reference's solver.py binds with ctypes (solver.py:92-98, 103, 131-134, 176-178)
 * from a library it calls fluid_wrapper.dll.  The source of that DLL is not part of the reference, so only the
 * signatures and the data layout solver.py relies on are mirrored; the behaviour below is this library's own:
 *   - one global simulation: resolution^3 cells, D3Q19 SRT FP32 with VOLUME_FORCE and UPDATE_FIELDS,
 *   - the STL is centred in the box and scaled so that its longest side is 0.9*resolution (solver.py:106-109),
 *   - pump_force acts along +z, all boundaries periodic,
 *   - velocities come back as three resolution^3 float32 arrays indexed n = x+(y+z*N)*N (solver.py:113-118, 140-142).
 * Built as cfd_amd/libfluid_wrapper.so from cfd_amd/host/fluid_wrapper.cpp.
 */
#ifndef FLUID_WRAPPER_H
#define FLUID_WRAPPER_H
#ifdef __cplusplus
extern "C" {
#endif
void fluid_init(int resolution, float viscosity, float pump_force, const char* stl_path); /* solver.py:95, 103 */
void fluid_step(int steps);                                                             /* solver.py:96, 131 */
void fluid_get_velocity(float* vx, float* vy, float* vz);                               /* solver.py:97-98, 134 */
void fluid_cleanup(void);                                                               /* solver.py:176-178 */
#ifdef __cplusplus
}
#endif
#endif
