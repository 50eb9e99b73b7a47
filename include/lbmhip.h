/*
 * lbmhip.h - C ABI of the MI355X lattice-Boltzmann time-step library (liblbmhip.so).
 *
 * This is the host<->device seam of the engine: what the reference's LBM_Domain reaches through its
 * Device / Memory<T> / Kernel wrapper (FluidX3D-master/src/opencl.hpp:280-689) is reached here through
 * plain C calls on an opaque domain handle.  One handle = one LBM_Domain = one lattice block on one GPU
 * (FluidX3D-master/src/lbm.hpp:20-204).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *  - every call returns 0 on success, non-zero on failure; lbmhip_last_error() gives the message of the
 *    last failure on the calling thread.  The reference prints and exits on every such error
 *    (FluidX3D-master/src/utilities.hpp:4159-4166); the C++ host above this ABI does the same.
 *  - one owner thread per handle; calls on one handle are not re-entrant.
 *  - launches are asynchronous on the handle's stream; lbmhip_sync() waits for it.
 *  - cell index n = x+(y+z*Ny)*Nx, DDFs fi[i*N+n], u[c*N+n], F[c*N+n] (FluidX3D-master/src/lbm.hpp:478-492,
 *    kernel.cpp:871-873, 970-972).  Nx, Ny, Nz are the LOCAL sizes including the 1-cell halo on every axis
 *    with more than one domain (FluidX3D-master/src/lbm.cpp:1059-1072).
 */
#ifndef LBMHIP_H
#define LBMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DDF storage (FluidX3D-master/src/lbm.cpp:561-609): arithmetic is always FP32 */
enum { LBMHIP_FP32 = 0, LBMHIP_FP16S = 1, LBMHIP_FP16C = 2 };
/* collision operator (FluidX3D-master/src/kernel.cpp:1613-1651) */
enum { LBMHIP_SRT = 0, LBMHIP_TRT = 1 };
/* extensions, the reference's compile-time switches of the same names (FluidX3D-master/src/defines.hpp.bak) */
enum {
	LBMHIP_UPDATE_FIELDS = 1,
	LBMHIP_VOLUME_FORCE = 2,
	LBMHIP_MOVING_BOUNDARIES = 4,
	LBMHIP_EQUILIBRIUM_BOUNDARIES = 8,
	LBMHIP_FORCE_FIELD = 16,
	LBMHIP_SUBGRID = 32
};
/* device fields of a domain (FluidX3D-master/src/lbm.hpp:38, 70-75) */
enum { LBMHIP_FIELD_FI = 0, LBMHIP_FIELD_RHO = 1, LBMHIP_FIELD_U = 2, LBMHIP_FIELD_FLAGS = 3, LBMHIP_FIELD_F = 4 };
/* what a halo exchange moves (FluidX3D-master/src/lbm.hpp:18, lbm.cpp:2184-2194) */
enum { LBMHIP_HALO_FI = 0, LBMHIP_HALO_RHO_U_FLAGS = 1, LBMHIP_HALO_FLAGS = 2, LBMHIP_HALO_F = 3 };
/* staging buffers of a domain: what it sends from its +/- face, what it receives for its +/- halo */
enum { LBMHIP_BUF_SEND_P = 0, LBMHIP_BUF_SEND_M = 1, LBMHIP_BUF_RECV_P = 2, LBMHIP_BUF_RECV_M = 3 };

typedef struct lbmhip_domain lbmhip_domain; /* opaque */

/* constructor arguments of LBM_Domain (FluidX3D-master/src/lbm.cpp:120-160) plus what the reference bakes
 * into the kernel text at run time (FluidX3D-master/src/lbm.cpp:424-665) */
typedef struct lbmhip_config {
	uint32_t Nx, Ny, Nz;   /* local lattice size including halo layers */
	uint32_t Dx, Dy, Dz;   /* number of domains per axis (halo on an axis iff D>1) */
	int32_t Ox, Oy, Oz;    /* offset of this domain in the global lattice */
	uint32_t velocity_set; /* 9 (D2Q9), 15, 19 or 27 */
	uint32_t storage;      /* LBMHIP_FP32 / FP16S / FP16C */
	uint32_t collision;    /* LBMHIP_SRT / LBMHIP_TRT */
	uint32_t features;     /* OR of the extension bits */
	float nu;              /* kinematic shear viscosity; relaxation rate w = 1/(3 nu + 1/2) */
	int32_t device;        /* HIP device ordinal */
} lbmhip_config;

/* ---- devices and errors: replaces get_devices() / print_error (opencl.hpp:234-241, utilities.hpp:4159) */
int lbmhip_device_count(int* count);
int lbmhip_device_name(int device, char* name, uint64_t capacity);
int lbmhip_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
const char* lbmhip_last_error(void);
const char* lbmhip_version(void);

/* ---- byte models: bytes_per_cell_device() / bandwidth_bytes_per_cell_device() (lbm.cpp:37-82) */
uint32_t lbmhip_bytes_per_cell_device(uint32_t velocity_set, uint32_t storage, uint32_t features);
uint32_t lbmhip_bandwidth_bytes_per_cell(uint32_t velocity_set, uint32_t storage, uint32_t features);

/* ---- domain life cycle: LBM_Domain::LBM_Domain + allocate() / ~LBM (lbm.cpp:120-229, 1125-1133).
 * Allocates fi, rho (=1), u (=0), flags (=0), F (=0, with FORCE_FIELD) and, when any D>1, the four staging
 * buffers.  Creates the handle's own stream. */
int lbmhip_domain_create(const lbmhip_config* config, lbmhip_domain** out);
int lbmhip_domain_destroy(lbmhip_domain* domain);
int lbmhip_domain_config(const lbmhip_domain* domain, lbmhip_config* out);

/* ---- streams: the in-order queue of Device (opencl.hpp:306, 330).  stream = a hipStream_t; NULL restores
 * the handle's own stream.  All following launches and copies of this handle go to that stream. */
int lbmhip_set_stream(lbmhip_domain* domain, void* stream);
int lbmhip_get_stream(lbmhip_domain* domain, void** stream);
int lbmhip_sync(lbmhip_domain* domain); /* Device::finish_queue (opencl.hpp:330) */
/* Second in-order queue of the handle for halo traffic (no counterpart in the reference, which runs pack, copy and
 * unpack on the one queue and blocks the host between them, lbm.cpp:2120-2182).
 * comm_begin: the communication stream waits for everything enqueued so far, and all following launches and copies
 * of this handle go to it; comm_end: back to the compute stream (no wait); comm_join: the compute stream waits for
 * everything enqueued so far on the communication stream. */
int lbmhip_comm_begin(lbmhip_domain* domain);
int lbmhip_comm_end(lbmhip_domain* domain);
int lbmhip_comm_join(lbmhip_domain* domain);

/* ---- host <-> device copies: Memory<T>::enqueue_write_to_device / enqueue_read_from_device with offset
 * and length (opencl.hpp:505-526, 603-606).  Offsets and sizes in bytes; the copy is complete when the call returns. */
int lbmhip_upload(lbmhip_domain* domain, int field, const void* host, uint64_t offset, uint64_t bytes);
int lbmhip_download(lbmhip_domain* domain, int field, void* host, uint64_t offset, uint64_t bytes);
/* Page-locked host memory for the host copies of the fields (no counterpart in the reference, whose Memory<T> allocates its
 * host buffer with new[], opencl.hpp:356-372): lbmhip_upload / lbmhip_download from and to such memory run at the full rate
 * of the host link (57 GB/s on MI355X against 17-29 GB/s for pageable memory).  Not tied to a device or a domain; returns
 * non-zero when the system refuses to lock that much memory, in which case the caller falls back to ordinary memory. */
int lbmhip_host_alloc(uint64_t bytes, void** host);
int lbmhip_host_free(void* host);
/* device address and size of a field, for callers that keep data resident (no copy) */
int lbmhip_field_pointer(lbmhip_domain* domain, int field, void** device_pointer, uint64_t* bytes);
/* Device layout of LBMHIP_FIELD_FI: stream i of the DDFs starts i*stride elements after the pointer above (element = 4 or
 * 2 bytes by storage).  stride >= Nx*Ny*Nz: lattices whose streams are longer than 64 MiB and would not start a multiple of 64 MiB apart (e.g. a slab with
 * halo planes) get 2^17 elements of padding between the streams, because such strides cost 1-4 % of the step rate on MI355X.  lbmhip_upload / lbmhip_download hide this: they
 * address the DDFs as Q streams of Nx*Ny*Nz elements back to back, the layout of the reference (kernel.cpp:871-873). */
int lbmhip_ddf_stride(const lbmhip_domain* domain, uint64_t* elements);

/* ---- kernels: LBM_Domain::enqueue_* (lbm.cpp:231-265, 305-311) */
int lbmhip_initialize(lbmhip_domain* domain);                                              /* kernel.cpp:1364-1446 */
int lbmhip_stream_collide(lbmhip_domain* domain, uint64_t t, float fx, float fy, float fz); /* kernel.cpp:1470-1654 */
/* the same step restricted to lattice planes z in [z_begin, z_end) (local coordinates); lets a caller run the
 * planes next to a halo first and overlap their exchange with the rest.  Cells are independent within a step. */
int lbmhip_stream_collide_planes(lbmhip_domain* domain, uint64_t t, float fx, float fy, float fz, uint32_t z_begin, uint32_t z_end);
/* `steps` consecutive steps t, t+1, ... with one force, nothing else happening in between: what LBM::run does per domain
 * when no exchange, force-field update or export falls between the steps (lbm.cpp:1354-1459, one stream_collide launch
 * per step there).  A domain that is not split in x or y, without UPDATE_FIELDS, may run them two per launch (same bits as
 * `steps` calls of lbmhip_stream_collide): band by band on each XCD, the second step a few lattice planes behind the first
 * and served by the caches instead of HBM.  Automatic on planes of at least 256 x 256 cells for the configurations with a
 * compiled two-step instance that was measured faster than a launch per step (collide_pairs.hip lists them: D3Q19 and D3Q27
 * without extensions in every DDF format, FP32 TRT, FP32 with SUBGRID, VOLUME_FORCE, EQUILIBRIUM_BOUNDARIES, FORCE_FIELD and their
 * usual combinations); every other configuration runs one launch per step unless "fuse" asks otherwise.  A paired launch never returns an error of its
 * own: should its assumptions about workgroup placement not hold, the launch that follows it finishes the work step by
 * step and the domain stops pairing (one line on stderr; lbmhip_get_tuning "band_given_up").  Tuning keys: lbmhip_set_tuning. */
int lbmhip_stream_collide_steps(lbmhip_domain* domain, uint64_t t, uint32_t steps, float fx, float fy, float fz);
/* Step t on the planes [z_begin, z_end) and step t+1 on the planes [z_begin+2, z_end-2), whose cells depend on nothing
 * outside [z_begin, z_end) (a slot and its owner are at most one plane apart, kernel.cpp:1332-1345): for the inner planes
 * of a z-slab, between the exchange of step t-1 and that of step t+1 - the caller steps the planes next to the halos by
 * lbmhip_stream_collide_planes around the exchange of step t.  One launch (plus the short one that finishes it) with the second
 * step trailing the first where lbmhip_stream_collide_steps would pair (same conditions; lbmhip_get_tuning "pairs_steps" tells
 * beforehand, "fused" afterwards), else two launches.  No counterpart
 * in the reference, which exchanges after every single launch (lbm.cpp:1411-1459). */
int lbmhip_stream_collide_pair_planes(lbmhip_domain* domain, uint64_t t, float fx, float fy, float fz, uint32_t z_begin, uint32_t z_end);
int lbmhip_update_fields(lbmhip_domain* domain, uint64_t t, float fx, float fy, float fz); /* kernel.cpp:1812-1894 */
int lbmhip_update_force_field(lbmhip_domain* domain, uint64_t t);                          /* kernel.cpp:1897-1910 */
int lbmhip_reset_force_field(lbmhip_domain* domain);                                       /* kernel.cpp:1911-1917 */
int lbmhip_update_moving_boundaries(lbmhip_domain* domain);                                /* kernel.cpp:1449-1465 */

/* ---- halo exchange: transfer_extract_* / transfer__insert_* (kernel.cpp:2130-2223) and
 * LBM_Domain::enqueue_transfer_extract_field / insert_field (lbm.cpp:2092-2119).  axis: 0=x, 1=y, 2=z.
 * pack fills SEND_P from the layer next to the + halo and SEND_M from the layer next to the - halo;
 * unpack writes RECV_P into the + halo layer and RECV_M into the - halo layer.  Buffer layout buf[b*A+a]. */
int lbmhip_halo_pack(lbmhip_domain* domain, int what, int axis, uint64_t t);
int lbmhip_halo_unpack(lbmhip_domain* domain, int what, int axis, uint64_t t);
int lbmhip_halo_bytes(const lbmhip_domain* domain, int what, int axis, uint64_t* bytes);
int lbmhip_halo_buffer(lbmhip_domain* domain, int which, void** device_pointer, uint64_t* capacity);
/* use caller-owned device memory (e.g. buffers registered with a communication library) as staging buffers */
int lbmhip_halo_set_buffers(lbmhip_domain* domain, void* send_p, void* send_m, void* recv_p, void* recv_m, uint64_t capacity);
/* the pointer swap of LBM::communicate_field (lbm.cpp:2129-2136) for domains that live in ONE process: copies
 * SEND_P of every domain into RECV_M of its + neighbour and SEND_M of that neighbour into RECV_P, device to
 * device (peer copies over xGMI when the domains sit on different GPUs), with periodic wrap.  domains[] is
 * ordered d = x+(y+z*Dy)*Dx.  Orders the copies after each domain's pack and before its unpack. */
int lbmhip_halo_exchange_local(lbmhip_domain** domains, uint32_t count, int what, int axis);
/* the same exchange (lbm.cpp:2129-2136) for ONE PROCESS PER GPU: sends SEND_P to rank_plus and SEND_M to rank_minus and
 * receives RECV_M from rank_minus and RECV_P from rank_plus, as one RCCL send/recv group on the handle's stream (ordered
 * after pack, before unpack; nothing blocks the host).  communicator is the caller's ncclComm_t for this GPU; the RCCL
 * entry points are looked up in the process at the first call (the caller that created the communicator has them loaded),
 * so the library itself does not link RCCL; the environment variable LBMHIP_RCCL_LIB names a particular library file instead.
 * rank_minus == rank_plus (two ranks, or one rank that is its own neighbour) works. */
int lbmhip_halo_exchange_rccl(lbmhip_domain* domain, void* communicator, int what, int axis, int rank_minus, int rank_plus);

/* ---- mesh voxelizer: LBM_Domain::voxelize_mesh_on_device / enqueue_unvoxelize_mesh_on_device (lbm.cpp:344-422) and the
 * kernels voxelize_mesh / unvoxelize_mesh (kernel.cpp:2295-2390).  p0, p1, p2: HOST arrays of 3*triangles floats (the three
 * corners of every triangle, in global lattice coordinates); bounding_box: x0 y0 z0 x1 y1 z1 of the mesh widened by 2
 * cells; direction: axis the rays travel along; t: the reference passes get_t()+1.  The velocity arguments may be NULL
 * (resting geometry).  Complete when the call returns. */
int lbmhip_voxelize_mesh(lbmhip_domain* domain, uint32_t direction, uint64_t t, uint8_t flag, const float* p0, const float* p1, const float* p2,
                         uint32_t triangles, const float* bounding_box, const float* rotation_center, const float* linear_velocity,
                         const float* rotational_velocity);
int lbmhip_unvoxelize_mesh(lbmhip_domain* domain, uint8_t flag, const float* bounding_box);
/* lbmhip_voxelize_mesh for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  Same arguments and results as lbmhip_voxelize_mesh, but the
 * triangles are staged in a fixed buffer inside the library's device image instead of a buffer owned by the domain: no
 * hipMalloc / hipFree, ever.  Meshes of more than LBMHIP_STATIC_MESH_TRIANGLES triangles are rejected with an error.  Each GPU
 * holds one such buffer (6.5 MiB) for the life of the process and every domain on that GPU shares it: the library lets calls on
 * one GPU take turns (a lock per GPU, held for the whole call), and a call only reads the triangles it has copied in itself.
 * Like lbmhip_voxelize_mesh it copies from the caller's host arrays and waits for the domain's stream before it returns, so it
 * cannot be used on a stream that is being captured into a graph. */
#define LBMHIP_STATIC_MESH_TRIANGLES 131072u
int lbmhip_voxelize_mesh_static(lbmhip_domain* domain, uint32_t direction, uint64_t t, uint8_t flag, const float* p0, const float* p1, const float* p2,
                                uint32_t triangles, const float* bounding_box, const float* rotation_center, const float* linear_velocity,
                                const float* rotational_velocity);

/* ---- sums over the cells whose flags EQUAL flag_marker: kernels object_center_of_mass / object_force / object_torque
 * (kernel.cpp:1929-1987) and LBM_Domain::enqueue_object_* (lbm.cpp:266-303).  what: 0 = positions (sums[0..2]) and cell
 * count (sums[3]), 1 = boundary force F, 2 = torque (position - center) x F.  force and torque use the F field as it
 * is: call lbmhip_update_force_field first.  Complete when the call returns.
 * global_size == NULL: the reference kernel's definition for this one domain - every cell of the array, halo layers
 * included, positions from the centre of the domain's own box (kernel.cpp:834-836).
 * global_size != NULL: halo layers are left to the domain that owns them and positions are measured from the centre of
 * the global box of global_size cells (lbm.hpp:501-503), so that adding the sums of all domains (lbm.cpp:1509-1554) gives
 * the object's true total; with one domain both definitions coincide.
 * The reference adds group sums with float atomics whose order is not fixed; here partial sums are added in a fixed order
 * in double, so the result is reproducible and agrees with the reference to float rounding of the summation. */
int lbmhip_object_sum(lbmhip_domain* domain, int what, uint8_t flag_marker, const float* center, const uint32_t* global_size, double* sums);

/* ---- DDF storage formats: the load / store conversions of the kernels (lbm.cpp:561-609: FP32 identity, FP16S IEEE half
 * scaled by 2^15, FP16C the custom 1-4-11 format of kernel.cpp:858-869; host-side counterparts utilities.hpp:121-144)
 * applied to a HOST array on `device`, with the same device code the lattice kernels use.  to_storage != 0: in = count
 * floats, out = count stored values (float for FP32, 16-bit otherwise); to_storage == 0: the reverse.  pairs != 0 converts
 * two neighbouring values at a time with the packed instructions the kernels use when a lane collides two cells together
 * (count must be even), pairs == 2 (16-bit formats) each half of a dword in place as the kernels do when a lane collides its two cells
 * one after the other; the results are the same.  For hosts that prepare or inspect raw DDFs, and for checking the
 * conversions exhaustively.  Complete when the call returns. */
int lbmhip_convert_ddfs(int device, uint32_t storage, int to_storage, int pairs, const void* in, void* out, uint64_t count);

/* ---- timing on the handle's stream (the reference only has a host clock, lbm.cpp:1481-1483) */
int lbmhip_timer_start(lbmhip_domain* domain);
int lbmhip_timer_stop(lbmhip_domain* domain, float* milliseconds); /* waits for the stream */

/* ---- tuning knobs (no counterpart in the reference): "cells_per_thread" (0 = auto, 1, 2, 4), "cells_together" (0 = auto,
 * 1 = a lane collides its cells one after the other, 2 = two at a time in packed FP32 arithmetic; only with 2 or 4 cells
 * per thread), "block" (0 = auto, 64..256 threads), "blocks_per_cu" (-1 = auto, 0 = no limit; caps the workgroups resident
 * per compute unit, counted in 256-lane workgroups), "autotune" (1 = default: while all these knobs are on auto, the first full-lattice
 * stream_collide launches of a lattice of at least 2^21 cells each try one launch geometry between two events (three
 * times each) and the one with the best median is kept; every geometry computes the same bits.  0 = fixed defaults; the environment variable
 * LBMHIP_AUTOTUNE=0 has the same effect for every domain created afterwards).
 * Two steps per launch (lbmhip_stream_collide_steps, lbmhip_stream_collide_pair_planes): "fuse" (-1 = automatic: the
 * configurations measured to gain, on planes of at least 256 x 256 cells, band by band; 0 = never; 1 = the form whose second step reads through the
 * Infinity Cache from any XCD, whenever a compiled instance exists; 2 = band by band on each XCD, whenever a compiled
 * instance exists; LBMHIP_FUSE=0|1|2 sets it for every domain created afterwards).  Band by band: "band_rows" (0 = automatic,
 * 2..4096 lattice rows per band), "band_lag" (0 = automatic, 2..64: planes the second step trails the first by),
 * "band_steps" (0 = automatic = 2; 2..4 time steps per launch of lbmhip_stream_collide_steps - more than two measured slower on
 * MI355X, see lbm_kernels.hpp), "band_blocks_per_cu" (-1 = automatic, 0 = no limit), "band_spin_limit" (-1 = default; polls before a wait gives up - 0 makes
 * every unfinished wait give up, for tests).  Form 1: "pair_lag" (0 = automatic, 3..64), "pair_blocks_per_cu".  Both:
 * "pair_cells_together" (16-bit DDFs: 1 or 2), "pair_block" (0 = automatic, 64..256 lanes per workgroup).
 * Read-only keys of lbmhip_get_tuning: "tuned" (the search has finished), "jit" (the last launch ran a kernel
 * compiled at run time), "pairs_steps" (two steps of this domain would share a launch), "fused" (what the last
 * lbmhip_stream_collide_steps / lbmhip_stream_collide_pair_planes call did: 0 = a launch per step, 1 or 2 = the form it
 * paired its steps in), "band_given_up" (paired launches found to have given up waits; their work was redone), "xcd_round_robin" (1 when
 * the probe at domain creation found workgroups i and i+8 of a launch on the same XCD, which the band-by-band form builds on; 0
 * sends "fuse" -1 and 2 to form 1). */
int lbmhip_set_tuning(lbmhip_domain* domain, const char* key, int value);
/* Reads back a launch setting ("cells_per_thread", "cells_together", "blocks_per_cu", "block", "autotune"), "tuned" (1 once
 * the geometry search has finished; the launch settings then report its choice) or "jit" (1 when the last stream_collide
 * ran a kernel compiled at run time).  No counterpart in the reference.
 * Kernel instances: the library carries compiled stream_collide instances for the common feature sets; for any other set the
 * first launch compiles an instance for exactly that set with hipRTC (2-4 s once per process and device), the way the
 * reference compiles its OpenCL code with the configured extensions (lbm.cpp:424-665).  Without hipRTC, or with
 * LBMHIP_JIT=0, an instance that tests the feature bits at run time is used (same results, 10-50 % slower). */
int lbmhip_get_tuning(lbmhip_domain* domain, const char* key, int* value);

#ifdef __cplusplus
}
#endif
#endif /* LBMHIP_H */
