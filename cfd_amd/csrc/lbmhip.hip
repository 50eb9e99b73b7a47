// lbmhip.hip - implementation of the C ABI declared in include/lbmhip.h (domain handle, launches, copies,
// halo staging).  Host code of a HIP translation unit; the heavy stream_collide instances live in collide_*.hip.
#include "../../include/lbmhip.h"
#define LBM_SMALL_KERNELS
#include "lbm_dispatch.hpp"
#include "lbm_jit.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

using namespace lbm;

// ---------------------------------------------------------------------------------------------- errors
static thread_local std::string g_error;
static int fail(const char* fmt, ...) {
	char buf[1024];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	g_error = buf;
	(void)hipGetLastError(); // HIP keeps the last error until it is read: a reported failure must not resurface in a later call's launch check
	return 1;
}
#define HIP_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(cond, ...) do { if (!(cond)) return fail(__VA_ARGS__); } while (0)

// ---------------------------------------------------------------------------------------------- domain
struct lbmhip_domain {
	lbmhip_config cfg;
	uint64_t N;
	uint64_t stride; // distance between two DDF streams in elements (N + padding)
	uint32_t fpxx_bytes;
	void* field[5];
	uint64_t field_bytes[5];
	hipStream_t own_stream, stream, comm_stream, saved_stream;
	hipEvent_t ev_fork, ev_join;
	void* halo_own[4];
	void* halo[4];
	uint64_t halo_capacity, halo_own_capacity; // of the buffers in use / of the library's own buffers
	hipEvent_t ev_start, ev_stop, ev_packed, ev_copied;
	int cells_per_thread, cells_together, block, blocks_per_cu;
	// launch-geometry search over the first full-lattice stream_collide launches (results do not depend on the geometry)
	struct { int V, P, blocks_per_cu, block; float ms[3]; } cand[64]; // three timed launches per candidate, the median decides
	int ncand, tune_launches, tune_pending, autotune, tuned, tuned_V, tuned_P, tuned_blocks_per_cu, tuned_block;
	hipEvent_t ev_tune0, ev_tune1;
	int uses_jit;  // the last stream_collide launch ran a kernel compiled at run time for this domain's feature set
	void* mesh; uint64_t mesh_capacity; // triangle corners of the last voxelized mesh (grown on demand)
	void* partial; // per-workgroup partial sums of the object_* reductions (kObjectSumBlocks float4)
	float w;
	// two time steps per launch (k_collide_pair): completion counters of the first step's workgroups
	uint32_t* pair_sync; uint64_t pair_sync_words; uint32_t pair_replica_stride;
	int fuse, pair_lag, pair_blocks_per_cu, pair_cells_together, pair_block; // fuse: -1 = automatic, 0 = never, 1 = whenever a compiled instance exists; pair_lag 0 = automatic; pair_blocks_per_cu -1 = automatic
	int pair_used, pair_used_lag, pair_used_blocks_per_cu; // what the last lbmhip_stream_collide_steps call ran (pair_used: 0 = a launch per step, 1 = k_collide_pair, 2 = k_collide_band)
	// two time steps per launch, band by band on each XCD (k_collide_band): 16 header words (error, redo count), the completion
	// counters, and the redo list; allocated at the first such launch
	uint32_t* band_sync; uint64_t band_sync_words; uint64_t* band_redo; uint64_t band_redo_capacity;
	int band_rows, band_lag, band_blocks_per_cu; // rows per band, lag in planes, residency cap: 0 / 0 / -1 = automatic
	int band_spin_limit, band_used_rows;
	int band_steps, band_used_steps;                // time steps per k_collide_band launch: 0 = automatic, 2..4
	uint32_t band_launches; // number of the last k_collide_band launch (the stamp its workgroups leave)
	int band_unchecked, band_given_up;            // launches since the error word was last read; launches found to have given up waits (their tasks were redone)
	int compute_units;
	int xcd_round_robin; // workgroups i and i+8 of a launch share an XCD (probed once per device)
};

// 64-bit safe fill (fields of the largest lattices exceed what 32-bit element counts can address)
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* p, uint32_t value, uint64_t count) {
	for (uint64_t i = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x; i < count; i += (uint64_t)gridDim.x*blockDim.x) p[i] = value;
}
static hipError_t fill_bytes(void* p, uint8_t byte, uint64_t bytes, hipStream_t stream) { // p is 256-byte aligned (hipMalloc)
	const uint32_t v = 0x01010101u*byte;
	const uint64_t words = bytes/4u;
	if (words) hipLaunchKernelGGL(k_fill_u32, dim3(4096), dim3(256), 0, stream, (uint32_t*)p, v, words);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && bytes%4u) e = hipMemsetAsync((char*)p+words*4u, byte, bytes%4u, stream);
	return e;
}
static hipError_t fill_words(void* p, uint32_t word, uint64_t count, hipStream_t stream) {
	if (count) hipLaunchKernelGGL(k_fill_u32, dim3(4096), dim3(256), 0, stream, (uint32_t*)p, word, count);
	return hipGetLastError();
}

// Distance between two DDF streams.  The streams of a lattice are read and written side by side, so the stride decides how
// the 2*Q concurrent streams fall onto the HBM channels.  Measured on MI355X (512 x 512 x Nz lattices): a stride that is a
// multiple of 64 MiB (512^3: 512 MiB; also 576, 768 MiB) is as good as it gets and padding it costs up to 2 %; any other
// stride is 1-4 % slower than that, worst for a slab with two halo planes (514 MiB), and 2^17 elements of padding per
// stream win it back (512x512x514 FP32: 34 070 -> 35 350 MLUPs/s, FP16S 70 000 -> 71 460; 600 planes: +2.2 %).  Streams
// shorter than 64 MiB (not measured) stay unpadded.  LBMHIP_PAD=<elements> overrides the choice for every domain created
// afterwards (diagnostic, tests).
static uint64_t ddf_stream_padding(const lbmhip_config* c, uint64_t N, uint32_t fpxx_bytes) {
	if (const char* pad = getenv("LBMHIP_PAD")) return (uint64_t)strtoull(pad, nullptr, 10);
	(void)c;
	if (N*fpxx_bytes < (64ull<<20) || (N*fpxx_bytes)%(64ull<<20) == 0ull) return 0ull;
	return 1ull<<17;
}

// Copies between the host's view of the DDFs (Q streams of N elements back to back) and the device layout (streams `stride` apart).
static hipError_t copy_ddfs(lbmhip_domain* d, void* host, uint64_t offset, uint64_t bytes, bool to_device) {
	const uint64_t stream_bytes = d->N*d->fpxx_bytes, stride_bytes = d->stride*d->fpxx_bytes;
	char* h = static_cast<char*>(host);
	char* dev = static_cast<char*>(d->field[LBMHIP_FIELD_FI]);
	while (bytes > 0u) {
		const uint64_t i = offset/stream_bytes, within = offset%stream_bytes, piece = std::min(bytes, stream_bytes-within);
		const hipError_t e = to_device ? hipMemcpyAsync(dev+i*stride_bytes+within, h, piece, hipMemcpyHostToDevice, d->stream)
		                               : hipMemcpyAsync(h, dev+i*stride_bytes+within, piece, hipMemcpyDeviceToHost, d->stream);
		if (e != hipSuccess) return e;
		h += piece; offset += piece; bytes -= piece;
	}
	return hipSuccess;
}

static const uint32_t kObjectSumBlocks = 2048u;
static const int kTuneReps = 3; // timed launches per candidate of the launch-geometry search
static uint32_t q_transfers(uint32_t Q) { return Q == 9u ? 3u : Q == 27u ? 9u : 5u; }
static uint32_t storage_bytes(uint32_t st) { return st == LBMHIP_FP32 ? 4u : 2u; }
static bool valid_q(uint32_t Q) { return Q == 9u || Q == 15u || Q == 19u || Q == 27u; }

struct DeviceGuard { // makes the domain's device current for the duration of a call
	int prev; bool ok;
	explicit DeviceGuard(int dev) : prev(0), ok(true) { (void)hipGetLastError(); /* drop a stale error left by other code in the process */ if (hipGetDevice(&prev) != hipSuccess || (prev != dev && hipSetDevice(dev) != hipSuccess)) ok = false; }
	~DeviceGuard() { int cur = 0; if (hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); }
};

static Args make_args(const lbmhip_domain* d, uint64_t t, float fx, float fy, float fz) {
	Args a;
	a.fi = d->field[LBMHIP_FIELD_FI]; a.rho = (float*)d->field[LBMHIP_FIELD_RHO]; a.u = (float*)d->field[LBMHIP_FIELD_U];
	a.flags = (uint8_t*)d->field[LBMHIP_FIELD_FLAGS]; a.F = (float*)d->field[LBMHIP_FIELD_F];
	a.N = d->N; a.S = d->stride; a.Nx = d->cfg.Nx; a.Ny = d->cfg.Ny; a.Nz = d->cfg.Nz;
	a.hx = d->cfg.Dx > 1u; a.hy = d->cfg.Dy > 1u; a.hz = d->cfg.Dz > 1u;
	a.z0 = a.hz; a.z1 = d->cfg.Nz-a.hz;
	a.odd = (uint32_t)(t&1ull); a.features = d->cfg.features; a.xchunks = 1u;
	a.fx = fx; a.fy = fy; a.fz = fz; a.w = d->w;
	a.tau0 = 1.0f/d->w; a.wm = 1.0f/(0.1875f/(1.0f/d->w-0.5f)+0.5f); // kernel.cpp:1598, 1624 (IEEE float division, as on the device)
	return a;
}

// RCCL entry points, resolved at first use: from the library LBMHIP_RCCL_LIB names, else from the running process (the
// caller's RCCL), else from librccl.so
namespace {
typedef int (*rccl_group_fn)(void);
typedef int (*rccl_p2p_fn)(void*, size_t, int, int, void*, hipStream_t); // ncclSend / ncclRecv (the send buffer is const in RCCL's prototype)
typedef const char* (*rccl_error_fn)(int);
struct RcclApi { rccl_group_fn group_start, group_end; rccl_p2p_fn send, recv; rccl_error_fn error_string; bool ok; };
const RcclApi& rccl_api() {
	static const RcclApi api = []() {
		RcclApi a = {};
		void* h = RTLD_DEFAULT;
		if (const char* path = getenv("LBMHIP_RCCL_LIB")) h = dlopen(path, RTLD_NOW|RTLD_LOCAL); // a particular build of the library
		else if (!dlsym(h, "ncclSend")) h = dlopen("librccl.so.1", RTLD_NOW|RTLD_GLOBAL);
		if (!h) h = dlopen("librccl.so", RTLD_NOW|RTLD_GLOBAL);
		if (!h) return a;
		a.group_start = (rccl_group_fn)dlsym(h, "ncclGroupStart"); a.group_end = (rccl_group_fn)dlsym(h, "ncclGroupEnd");
		a.send = (rccl_p2p_fn)dlsym(h, "ncclSend"); a.recv = (rccl_p2p_fn)dlsym(h, "ncclRecv");
		a.error_string = (rccl_error_fn)dlsym(h, "ncclGetErrorString");
		a.ok = a.group_start && a.group_end && a.send && a.recv;
		return a;
	}();
	return api;
}
} // namespace

extern "C" {

const char* lbmhip_last_error(void) { return g_error.c_str(); }
const char* lbmhip_version(void) { return "lbmhip 0.1 (gfx950)"; }

int lbmhip_device_count(int* count) {
	NEED(count, "lbmhip_device_count: null argument");
	int n = 0;
	const hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) { *count = 0; return fail("hipGetDeviceCount failed: %s", hipGetErrorString(e)); }
	*count = n;
	return 0;
}
int lbmhip_device_name(int device, char* name, uint64_t capacity) {
	NEED(name && capacity > 0u, "lbmhip_device_name: null argument");
	hipDeviceProp_t p;
	HIP_TRY(hipGetDeviceProperties(&p, device));
	snprintf(name, (size_t)capacity, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
	return 0;
}
int lbmhip_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
	DeviceGuard g(device);
	NEED(g.ok, "cannot select device %d", device);
	size_t f = 0, t = 0;
	HIP_TRY(hipMemGetInfo(&f, &t));
	if (free_bytes) *free_bytes = f;
	if (total_bytes) *total_bytes = t;
	return 0;
}

// FluidX3D-master/src/lbm.cpp:37-50
uint32_t lbmhip_bytes_per_cell_device(uint32_t Q, uint32_t storage, uint32_t features) {
	uint32_t b = Q*storage_bytes(storage)+17u;
	if (features&LBMHIP_FORCE_FIELD) b += 12u;
	return b;
}
// FluidX3D-master/src/lbm.cpp:51-82
uint32_t lbmhip_bandwidth_bytes_per_cell(uint32_t Q, uint32_t storage, uint32_t features) {
	uint32_t b = Q*2u*storage_bytes(storage)+1u;
	if (features&LBMHIP_UPDATE_FIELDS) b += 16u;
	if (features&LBMHIP_FORCE_FIELD) b += 12u;
	if (features&LBMHIP_MOVING_BOUNDARIES) b += Q-1u;
	return b;
}

// Which XCD the workgroups of a launch land on.  k_collide_band lets blockIdx.x % 8 stand for "the same XCD" (workgroups are
// dealt to the XCDs round-robin by their linear index; its grids are multiples of 8 wide).  The launch itself checks every
// hand-over against the XCD number it reads from the hardware, so a false assumption only costs time; this probe, once per
// device, finds it out before the first such launch: 16 x 4 workgroups record where they ran, and every workgroup has to
// sit where the first workgroup with the same blockIdx.x % 8 sits.
__global__ __launch_bounds__(64) void k_probe_xcd(uint32_t* where) {
	if (threadIdx.x == 0u) where[blockIdx.y*gridDim.x+blockIdx.x] = xcd_of_this_wave();
}
static int xcd_round_robin(int device) { // 1: as assumed, 0: not (or the probe failed); the device is current
	static std::mutex lock;
	static std::map<int, int> known;
	std::lock_guard<std::mutex> guard(lock);
	const auto it = known.find(device);
	if (it != known.end()) return it->second;
	int result = 0;
	uint32_t* where = nullptr;
	uint32_t host[64];
	if (hipMalloc((void**)&where, sizeof(host)) == hipSuccess) {
		hipLaunchKernelGGL(k_probe_xcd, dim3(16, 4), dim3(64), 0, nullptr, where);
		if (hipGetLastError() == hipSuccess && hipMemcpy(host, where, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
			result = 1;
			for (int k = 0; k < 64; k++) if (host[k] != host[k%8]) result = 0;
		}
		(void)hipFree(where);
	}
	(void)hipGetLastError();
	if (!result) fprintf(stderr, "lbmhip: device %d does not place workgroups on its XCDs round-robin; two steps per launch use the form that does not depend on it\n", device);
	known[device] = result;
	return result;
}


int lbmhip_domain_create(const lbmhip_config* c, lbmhip_domain** out) {
	NEED(c && out, "lbmhip_domain_create: null argument");
	*out = nullptr;
	NEED((uint64_t)c->Nx*c->Ny*c->Nz > 0u, "Grid point number is 0: %ux%ux%u = 0.", c->Nx, c->Ny, c->Nz);
	NEED(c->Dx*c->Dy*c->Dz > 0u, "You specified 0 LBM grid domains (%ux%ux%u).", c->Dx, c->Dy, c->Dz);
	NEED(valid_q(c->velocity_set), "unsupported velocity set %u (use 9, 15, 19 or 27)", c->velocity_set);
	NEED(c->velocity_set != 9u || c->Nz == 1u, "D2Q9 is the 2D velocity set. You have to set Nz=1u.");
	NEED(c->storage <= LBMHIP_FP16C, "unknown DDF storage %u", c->storage);
	NEED(c->collision <= LBMHIP_TRT, "No LBM collision operator selected (collision=%u).", c->collision);
	NEED(c->nu > 0.0f, c->nu == 0.0f ? "Viscosity cannot be 0." : "Viscosity cannot be negative.");
	NEED((c->features&~63u) == 0u, "unknown feature bits 0x%x", c->features);
	NEED(c->Nx < (1u<<30), "lattice rows of %u cells exceed the 2^30 limit", c->Nx);
	NEED(c->Nz <= 65535u && (uint64_t)c->Nx*c->Ny < 0xFFFFFF00ull, "lattice %ux%ux%u exceeds the launch limits (Nz <= 65535, Nx*Ny < 2^32)", c->Nx, c->Ny, c->Nz);
	for (int ax = 0; ax < 3; ax++) {
		const uint32_t n = ax == 0 ? c->Nx : ax == 1 ? c->Ny : c->Nz, dd = ax == 0 ? c->Dx : ax == 1 ? c->Dy : c->Dz;
		NEED(dd <= 1u || n >= 3u, "an axis with %u domains needs at least 1 cell between its two halo layers (size %u)", dd, n);
	}
	DeviceGuard g(c->device);
	NEED(g.ok, "cannot select device %d", c->device);

	lbmhip_domain* d = new lbmhip_domain();
	memset(d, 0, sizeof(*d));
	d->cfg = *c;
	d->N = (uint64_t)c->Nx*c->Ny*c->Nz;
	d->fpxx_bytes = storage_bytes(c->storage);
	d->w = 1.0f/(3.0f*c->nu+0.5f); // relaxation rate, float like LBM_Domain::get_tau (lbm.hpp:132, lbm.cpp:509-511)
	d->cells_per_thread = 0; // 0 = automatic
	d->cells_together = 0;
	d->block = 0;
	d->blocks_per_cu = -1; // -1 = automatic
	d->tune_pending = -1;
	{ const char* env = getenv("LBMHIP_AUTOTUNE"); d->autotune = !(env && env[0] == '0'); }
	d->stride = d->N+ddf_stream_padding(c, d->N, d->fpxx_bytes);
	d->field_bytes[LBMHIP_FIELD_FI] = d->N*c->velocity_set*d->fpxx_bytes; // what uploads and downloads see: the streams back to back
	d->field_bytes[LBMHIP_FIELD_RHO] = d->N*4u;
	d->field_bytes[LBMHIP_FIELD_U] = d->N*12u;
	d->field_bytes[LBMHIP_FIELD_FLAGS] = d->N;
	d->field_bytes[LBMHIP_FIELD_F] = (c->features&LBMHIP_FORCE_FIELD) ? d->N*12u : 0u;

	auto cleanup = [&]() { lbmhip_domain_destroy(d); };
	hipError_t e = hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking);
	if (e != hipSuccess) { cleanup(); return fail("hipStreamCreate failed: %s", hipGetErrorString(e)); }
	d->stream = d->own_stream;
	for (int k = 0; k < 5; k++) {
		if (d->field_bytes[k] == 0u) continue;
		e = hipMalloc(&d->field[k], k == LBMHIP_FIELD_FI ? d->stride*c->velocity_set*d->fpxx_bytes : d->field_bytes[k]);
		if (e != hipSuccess) {
			const uint64_t need = d->N*lbmhip_bytes_per_cell_device(c->velocity_set, c->storage, c->features);
			cleanup();
			return fail("Grid resolution (%u, %u, %u) is too large: %llu MB required on device %d (%s).", c->Nx, c->Ny, c->Nz,
			            (unsigned long long)(need/1048576ull), c->device, hipGetErrorString(e));
		}
	}
	// defaults of LBM_Domain::allocate (lbm.cpp:164-167, 175): rho = 1, everything else 0; fi is left to initialize()
	e = fill_bytes(d->field[LBMHIP_FIELD_FI], 0u, d->stride*c->velocity_set*d->fpxx_bytes, d->stream);
	if (e == hipSuccess) e = fill_words(d->field[LBMHIP_FIELD_RHO], 0x3F800000u, d->N, d->stream);
	if (e == hipSuccess) e = fill_bytes(d->field[LBMHIP_FIELD_U], 0u, d->field_bytes[LBMHIP_FIELD_U], d->stream);
	if (e == hipSuccess) e = fill_bytes(d->field[LBMHIP_FIELD_FLAGS], 0u, d->field_bytes[LBMHIP_FIELD_FLAGS], d->stream);
	if (e == hipSuccess && d->field[LBMHIP_FIELD_F]) e = fill_bytes(d->field[LBMHIP_FIELD_F], 0u, d->field_bytes[LBMHIP_FIELD_F], d->stream);
	if (e != hipSuccess) { cleanup(); return fail("clearing the fields failed: %s", hipGetErrorString(e)); }

	if (c->Dx*c->Dy*c->Dz > 1u) { // LBM_Domain::allocate_transfer (lbm.cpp:2019-2035): one set of buffers sized for the largest face
		uint64_t amax = 0u;
		if (c->Dx > 1u) amax = std::max<uint64_t>(amax, (uint64_t)c->Ny*c->Nz);
		if (c->Dy > 1u) amax = std::max<uint64_t>(amax, (uint64_t)c->Nz*c->Nx);
		if (c->Dz > 1u) amax = std::max<uint64_t>(amax, (uint64_t)c->Nx*c->Ny);
		d->halo_capacity = d->halo_own_capacity = amax*std::max<uint32_t>(q_transfers(c->velocity_set)*d->fpxx_bytes, 17u);
		for (int k = 0; k < 4; k++) {
			e = hipMalloc(&d->halo_own[k], d->halo_capacity);
			if (e != hipSuccess) { cleanup(); return fail("allocating halo staging buffers failed: %s", hipGetErrorString(e)); }
			d->halo[k] = d->halo_own[k];
		}
	}
	e = hipMalloc(&d->partial, kObjectSumBlocks*4u*sizeof(float));
	d->fuse = -1; d->pair_lag = 0; d->pair_blocks_per_cu = -1; d->pair_cells_together = 1;
	d->band_rows = 0; d->band_lag = 0; d->band_blocks_per_cu = -1; d->band_spin_limit = -1; d->band_steps = 0; d->band_used_steps = 0;
	{ const char* env = getenv("LBMHIP_FUSE"); if (env && env[0] >= '0' && env[0] <= '2' && !env[1]) d->fuse = env[0]-'0'; }
	if (e == hipSuccess) e = hipDeviceGetAttribute(&d->compute_units, hipDeviceAttributeMultiprocessorCount, c->device);
	d->xcd_round_robin = e == hipSuccess && c->Dx == 1u && c->Dy == 1u && c->Nz >= 8u ? xcd_round_robin(c->device) : 0;
	if (e == hipSuccess && c->Dx == 1u && c->Dy == 1u && c->Nz >= 8u) { // what k_collide_pair needs, whether or not it is ever used (264 KB at 512 planes)
		d->pair_replica_stride = (c->Nz+63u)/64u*64u+32u; // the replicas start on different cache lines, an odd number of lines apart
		d->pair_sync_words = (uint64_t)PAIR_REPLICAS*d->pair_replica_stride+(uint64_t)c->Nz*PAIR_SHARDS*PAIR_SHARD_STRIDE;
		e = hipMalloc((void**)&d->pair_sync, d->pair_sync_words*sizeof(uint32_t));
		if (e == hipSuccess) e = hipMemsetAsync(d->pair_sync, 0, d->pair_sync_words*sizeof(uint32_t), d->stream);
	}
	if (e == hipSuccess) e = hipEventCreate(&d->ev_start);
	if (e == hipSuccess) e = hipEventCreate(&d->ev_stop);
	if (e == hipSuccess) e = hipEventCreate(&d->ev_tune0);
	if (e == hipSuccess) e = hipEventCreate(&d->ev_tune1);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_packed, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_copied, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
	if (e != hipSuccess) { cleanup(); return fail("domain setup failed: %s", hipGetErrorString(e)); }
	*out = d;
	return 0;
}

int lbmhip_domain_destroy(lbmhip_domain* d) {
	if (!d) return 0;
	DeviceGuard g(d->cfg.device);
	if (d->own_stream) (void)hipStreamSynchronize(d->own_stream);
	for (int k = 0; k < 5; k++) if (d->field[k]) (void)hipFree(d->field[k]);
	for (int k = 0; k < 4; k++) if (d->halo_own[k]) (void)hipFree(d->halo_own[k]);
	if (d->partial) (void)hipFree(d->partial);
	if (d->pair_sync) (void)hipFree(d->pair_sync);
	if (d->band_sync) (void)hipFree(d->band_sync);
	if (d->band_redo) (void)hipFree(d->band_redo);
	if (d->mesh) (void)hipFree(d->mesh);
	if (d->ev_start) (void)hipEventDestroy(d->ev_start);
	if (d->ev_stop) (void)hipEventDestroy(d->ev_stop);
	if (d->ev_tune0) (void)hipEventDestroy(d->ev_tune0);
	if (d->ev_tune1) (void)hipEventDestroy(d->ev_tune1);
	if (d->ev_packed) (void)hipEventDestroy(d->ev_packed);
	if (d->ev_copied) (void)hipEventDestroy(d->ev_copied);
	if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
	if (d->ev_join) (void)hipEventDestroy(d->ev_join);
	if (d->comm_stream) { (void)hipStreamSynchronize(d->comm_stream); (void)hipStreamDestroy(d->comm_stream); }
	if (d->own_stream) (void)hipStreamDestroy(d->own_stream);
	delete d;
	return 0;
}
int lbmhip_domain_config(const lbmhip_domain* d, lbmhip_config* out) {
	NEED(d && out, "lbmhip_domain_config: null argument");
	*out = d->cfg;
	return 0;
}

int lbmhip_set_stream(lbmhip_domain* d, void* stream) {
	NEED(d, "null domain");
	d->stream = stream ? (hipStream_t)stream : d->own_stream;
	return 0;
}
int lbmhip_get_stream(lbmhip_domain* d, void** stream) {
	NEED(d && stream, "null argument");
	*stream = (void*)d->stream;
	return 0;
}
// After the stream has been synchronised: did a wait inside a two-step launch run out?  (It does not while workgroups are
// dealt to the XCDs round-robin and dispatched in grid order, which HIP does not promise.)  The tasks of such waits were
// done again by the launch behind, so this costs time only: the domain stops pairing its steps and says so once.
static int pair_check(lbmhip_domain* d) {
	if (!d->band_unchecked || !d->band_sync) return 0;
	uint32_t flag = 0u;
	HIP_TRY(hipMemcpy(&flag, d->band_sync, sizeof(flag), hipMemcpyDeviceToHost));
	d->band_unchecked = 0;
	if (flag) { // the lattice is right (k_collide_seam ran what was given up), the time is lost
		d->band_given_up++;
		d->fuse = 0;
		fprintf(stderr, "lbmhip: a two-step launch gave up waiting for its first step and finished step by step; two-step launches are now off for this domain\n");
	}
	return 0;
}
int lbmhip_sync(lbmhip_domain* d) {
	NEED(d, "null domain");
	DeviceGuard g(d->cfg.device);
	HIP_TRY(hipStreamSynchronize(d->stream));
	if (d->saved_stream) HIP_TRY(hipStreamSynchronize(d->saved_stream));
	if (d->comm_stream != d->stream) HIP_TRY(hipStreamSynchronize(d->comm_stream));
	return pair_check(d);
}
int lbmhip_comm_begin(lbmhip_domain* d) {
	NEED(d, "null domain");
	NEED(!d->saved_stream, "lbmhip_comm_begin: already on the communication stream");
	DeviceGuard g(d->cfg.device);
	HIP_TRY(hipEventRecord(d->ev_fork, d->stream));
	HIP_TRY(hipStreamWaitEvent(d->comm_stream, d->ev_fork, 0));
	d->saved_stream = d->stream;
	d->stream = d->comm_stream;
	return 0;
}
int lbmhip_comm_end(lbmhip_domain* d) {
	NEED(d, "null domain");
	NEED(d->saved_stream, "lbmhip_comm_end without lbmhip_comm_begin");
	d->stream = d->saved_stream;
	d->saved_stream = nullptr;
	return 0;
}
int lbmhip_comm_join(lbmhip_domain* d) {
	NEED(d, "null domain");
	NEED(!d->saved_stream, "lbmhip_comm_join: call lbmhip_comm_end first");
	DeviceGuard g(d->cfg.device);
	HIP_TRY(hipEventRecord(d->ev_join, d->comm_stream));
	HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_join, 0));
	return 0;
}

int lbmhip_host_alloc(uint64_t bytes, void** host) {
	NEED(host, "null argument");
	*host = nullptr;
	NEED(bytes > 0u, "lbmhip_host_alloc of 0 bytes");
	void* p = nullptr;
	const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable);
	if (e != hipSuccess || !p) return fail("page-locked host memory of %llu bytes refused: %s", (unsigned long long)bytes, hipGetErrorString(e));
	*host = p;
	return 0;
}
int lbmhip_host_free(void* host) {
	if (!host) return 0;
	HIP_TRY(hipHostFree(host));
	return 0;
}
int lbmhip_upload(lbmhip_domain* d, int field, const void* host, uint64_t offset, uint64_t bytes) {
	NEED(d && host, "null argument");
	NEED(field >= 0 && field < 5 && d->field[field], "field %d does not exist in this domain", field);
	NEED(bytes <= d->field_bytes[field] && offset <= d->field_bytes[field]-bytes, "upload of %llu bytes at offset %llu exceeds field %d (%llu bytes)",
	     (unsigned long long)bytes, (unsigned long long)offset, field, (unsigned long long)d->field_bytes[field]);
	DeviceGuard g(d->cfg.device);
	if (field == LBMHIP_FIELD_FI && d->stride != d->N) HIP_TRY(copy_ddfs(d, const_cast<void*>(host), offset, bytes, true));
	else HIP_TRY(hipMemcpyAsync((char*)d->field[field]+offset, host, bytes, hipMemcpyHostToDevice, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	return 0;
}
int lbmhip_download(lbmhip_domain* d, int field, void* host, uint64_t offset, uint64_t bytes) {
	NEED(d && host, "null argument");
	NEED(field >= 0 && field < 5 && d->field[field], "field %d does not exist in this domain", field);
	NEED(bytes <= d->field_bytes[field] && offset <= d->field_bytes[field]-bytes, "download of %llu bytes at offset %llu exceeds field %d (%llu bytes)",
	     (unsigned long long)bytes, (unsigned long long)offset, field, (unsigned long long)d->field_bytes[field]);
	DeviceGuard g(d->cfg.device);
	if (field == LBMHIP_FIELD_FI && d->stride != d->N) HIP_TRY(copy_ddfs(d, host, offset, bytes, false));
	else HIP_TRY(hipMemcpyAsync(host, (const char*)d->field[field]+offset, bytes, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	return pair_check(d);
}
int lbmhip_field_pointer(lbmhip_domain* d, int field, void** p, uint64_t* bytes) {
	NEED(d && p, "null argument");
	NEED(field >= 0 && field < 5 && d->field[field], "field %d does not exist in this domain", field);
	*p = d->field[field];
	if (bytes) *bytes = field == LBMHIP_FIELD_FI ? d->stride*d->cfg.velocity_set*d->fpxx_bytes : d->field_bytes[field];
	return 0;
}
int lbmhip_ddf_stride(const lbmhip_domain* d, uint64_t* elements) {
	NEED(d && elements, "null argument");
	*elements = d->stride;
	return 0;
}

static int pairs_steps(const lbmhip_domain* d);
int lbmhip_set_tuning(lbmhip_domain* d, const char* key, int value) {
	NEED(d && key, "null argument");
	if (!strcmp(key, "cells_per_thread")) { NEED(value == 0 || value == 1 || value == 2 || value == 4, "cells_per_thread must be 0 (auto), 1, 2 or 4"); d->cells_per_thread = value; return 0; }
	if (!strcmp(key, "cells_together")) { NEED(value == 0 || value == 1 || value == 2, "cells_together must be 0 (auto), 1 or 2"); d->cells_together = value; return 0; }
	if (!strcmp(key, "blocks_per_cu")) { NEED(value >= -1 && value <= 32, "blocks_per_cu must be -1 (auto), 0 (no limit) or 1..32"); d->blocks_per_cu = value; return 0; }
	if (!strcmp(key, "block")) { NEED(value == 0 || (value >= 64 && value <= 256 && value%64 == 0), "block must be 0 (auto), 64, 128, 192 or 256"); d->block = value; return 0; }
	if (!strcmp(key, "autotune")) { NEED(value == 0 || value == 1, "autotune must be 0 or 1"); d->autotune = value; if (!value) { d->tuned = 0; d->tune_launches = 0; d->tune_pending = -1; } return 0; }
	if (!strcmp(key, "fuse")) { NEED(value >= -1 && value <= 2, "fuse must be -1 (auto), 0 (never), 1 (k_collide_pair) or 2 (k_collide_band)"); d->fuse = value; return 0; }
	if (!strcmp(key, "band_rows")) { NEED(value == 0 || (value >= 2 && value <= 4096), "band_rows must be 0 (auto) or 2..4096"); d->band_rows = value; return 0; }
	if (!strcmp(key, "band_lag")) { NEED(value == 0 || (value >= 2 && value <= 64), "band_lag must be 0 (auto) or 2..64"); d->band_lag = value; return 0; }
	if (!strcmp(key, "band_blocks_per_cu")) { NEED(value >= -1 && value <= 32, "band_blocks_per_cu must be -1 (auto), 0 (no limit) or 1..32"); d->band_blocks_per_cu = value; return 0; }
	if (!strcmp(key, "band_steps")) { NEED(value == 0 || (value >= 2 && value <= 4), "band_steps must be 0 (auto) or 2..4"); d->band_steps = value; return 0; }
	if (!strcmp(key, "band_spin_limit")) { NEED(value >= -1, "band_spin_limit must be -1 (default) or a number of polls"); d->band_spin_limit = value; return 0; }
	if (!strcmp(key, "pair_lag")) { NEED(value == 0 || (value >= 3 && value <= 64), "pair_lag must be 0 (auto) or 3..64"); d->pair_lag = value; return 0; }
	if (!strcmp(key, "pair_cells_together")) { NEED(value == 1 || value == 2, "pair_cells_together must be 1 or 2"); d->pair_cells_together = value; return 0; }
	if (!strcmp(key, "pair_block")) { NEED(value == 0 || (value >= 64 && value <= 256 && value%64 == 0), "pair_block must be 0 (auto), 64, 128, 192 or 256"); d->pair_block = value; return 0; }
	if (!strcmp(key, "pair_blocks_per_cu")) { NEED(value >= -1 && value <= 32, "pair_blocks_per_cu must be -1 (auto), 0 (no limit) or 1..32"); d->pair_blocks_per_cu = value; return 0; }
	return fail("unknown tuning key '%s'", key);
}
int lbmhip_get_tuning(lbmhip_domain* d, const char* key, int* value) {
	NEED(d && key && value, "null argument");
	if (!strcmp(key, "cells_per_thread")) { *value = d->tuned ? d->tuned_V : d->cells_per_thread; return 0; }
	if (!strcmp(key, "cells_together")) { *value = d->tuned ? d->tuned_P : d->cells_together; return 0; }
	if (!strcmp(key, "blocks_per_cu")) { *value = d->tuned ? d->tuned_blocks_per_cu : d->blocks_per_cu; return 0; }
	if (!strcmp(key, "block")) { *value = d->tuned ? d->tuned_block : d->block; return 0; }
	if (!strcmp(key, "autotune")) { *value = d->autotune; return 0; }
	if (!strcmp(key, "tuned")) { *value = d->tuned; return 0; }
	if (!strcmp(key, "jit")) { *value = d->uses_jit; return 0; }
	if (!strcmp(key, "fuse")) { *value = d->fuse; return 0; }
	if (!strcmp(key, "fused")) { *value = d->pair_used; return 0; } // the last lbmhip_stream_collide_steps call ran two steps per launch (1: k_collide_pair, 2: k_collide_band)
	if (!strcmp(key, "band_rows")) { *value = d->pair_used == 2 ? d->band_used_rows : d->band_rows; return 0; }
	if (!strcmp(key, "band_lag")) { *value = d->pair_used == 2 ? d->pair_used_lag : d->band_lag; return 0; }
	if (!strcmp(key, "band_blocks_per_cu")) { *value = d->pair_used == 2 ? d->pair_used_blocks_per_cu : d->band_blocks_per_cu; return 0; }
	if (!strcmp(key, "band_steps")) { *value = d->pair_used == 2 ? d->band_used_steps : d->band_steps; return 0; }
	if (!strcmp(key, "band_given_up")) { *value = d->band_given_up; return 0; } // two-step launches whose waits ran out (checked when the stream is synchronised)
	if (!strcmp(key, "pair_lag")) { *value = d->pair_used ? d->pair_used_lag : d->pair_lag; return 0; }
	if (!strcmp(key, "pair_cells_together")) { *value = d->pair_cells_together; return 0; }
	if (!strcmp(key, "pair_block")) { *value = d->pair_block; return 0; }
	if (!strcmp(key, "xcd_round_robin")) { *value = d->xcd_round_robin; return 0; } // what the placement probe found (0 also for domains that never pair)
	if (!strcmp(key, "pairs_steps")) { *value = pairs_steps(d); return 0; } // would two steps of this domain share a launch (whole lattice, or the inner planes of a z-slab)?
	if (!strcmp(key, "pair_blocks_per_cu")) { *value = d->pair_used ? d->pair_used_blocks_per_cu : d->pair_blocks_per_cu; return 0; }
	return fail("unknown tuning key '%s'", key);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- launches
// A stream_collide / update_fields instance: precompiled (fn), or compiled at run time for exactly this feature set (jit)
struct CollideKernel { collide_launch_fn fn; hipFunction_t jit; };
static CollideKernel find_collide(const lbmhip_domain* d, int V, int P, bool fields_only) {
	const int Q = (int)d->cfg.velocity_set, ST = (int)d->cfg.storage, COLL = (int)d->cfg.collision;
	const bool small = Q == 9 || Q == 15;
	collide_launch_fn fn = nullptr;
	for (int pass = 0; pass < 3 && !fn; pass++) { // 0: precompiled for exactly this feature set; 1: compiled now for it; 2: run-time features
		if (pass < 2 && (fields_only || getenv("LBMHIP_GENERIC"))) continue; // LBMHIP_GENERIC: diagnostic, measures the run-time-feature instances
		if (pass == 1) {
			const lbm::jit::Instance* inst = lbm::jit::instance(d->cfg.device, Q, ST, COLL, (int)d->cfg.features, small);
			hipFunction_t f = inst ? inst->fn[lbm::jit::geometry_index(small ? 1 : V, P)] : nullptr;
			if (f) return CollideKernel{nullptr, f};
			continue;
		}
		const int FEAT = pass == 0 ? (int)d->cfg.features : -1;
		if (Q == 19) fn = ST == ST_FP32 ? select_collide_q19_fp32(COLL, FEAT, V, P, fields_only) : ST == ST_FP16S ? select_collide_q19_fp16s(COLL, FEAT, V, P, fields_only) : select_collide_q19_fp16c(COLL, FEAT, V, P, fields_only);
		else if (Q == 27) fn = ST == ST_FP32 ? select_collide_q27_fp32(COLL, FEAT, V, P, fields_only) : ST == ST_FP16S ? select_collide_q27_fp16s(COLL, FEAT, V, P, fields_only) : select_collide_q27_fp16c(COLL, FEAT, V, P, fields_only);
		else fn = select_collide_small(Q, ST, COLL, FEAT, V, P, fields_only);
	}
	return CollideKernel{fn, nullptr};
}

// Default launch geometry.  Measured on MI355X at 512^3: plain FP32 D3Q19 is fastest with one cell per lane (dword
// accesses), 256-lane workgroups and at most 12 waves resident per CU (more waves in flight lower the achieved HBM
// bandwidth by ~15 %); the 16-bit formats with the 4-cells-per-lane kernel (aligned 8-byte accesses); rows that are not a
// multiple of 4 cells fall back to 2 or 1 cells per lane.  Heavier collision variants (TRT, subgrid,
// field output) want more waves, which the search below finds on the running lattice.
static void default_geometry(const lbmhip_domain* d, bool fields_only, int* V_out, int* P_out, int* blocks_per_cu_out) {
	const bool small = d->cfg.velocity_set == 9u || d->cfg.velocity_set == 15u;
	const bool half = d->cfg.storage != LBMHIP_FP32;
	int V = d->cells_per_thread;
	if (V == 0 && fields_only) V = half ? 4 : 1; // update_fields at 512^3: FP32 70 % of 8 TB/s with dwords (64 % with 8-byte loads), 16-bit DDFs 69 % with 4 cells per lane (64 % with 1)
	if (V == 0) V = half ? 4 : d->cfg.velocity_set == 27u ? 2 : 1; // D3Q27 FP32: 27 dword streams per lane leave too little in flight
	if (small) V = 1;
	while (V > 1 && d->cfg.Nx%(uint32_t)V != 0u) V /= 2; // vector accesses need rows that are whole vectors
	int blocks_per_cu = d->blocks_per_cu;
	if (blocks_per_cu < 0) {
		if (fields_only) blocks_per_cu = half ? 0 : 6; // FP32: 70.6 % with at most 24 waves per CU against 67.9 % without a cap
		else if (!half) blocks_per_cu = d->cfg.velocity_set == 27u ? 4 : 3;
		else blocks_per_cu = (d->cfg.storage == LBMHIP_FP16S && d->cfg.velocity_set == 19u && V == 4) ? 7 : 0;
	}
	*V_out = V; *P_out = (V > 1 && d->cells_together == 2) ? 2 : 1; *blocks_per_cu_out = blocks_per_cu;
}

// the geometries worth trying for this domain (blocks_per_cu counts 256-lane workgroups)
static void build_candidates(lbmhip_domain* d) {
	const bool small = d->cfg.velocity_set == 9u || d->cfg.velocity_set == 15u;
	const bool half = d->cfg.storage != LBMHIP_FP32;
	d->ncand = 0;
	auto add = [&](int V, int P, std::initializer_list<int> caps, std::initializer_list<int> ragged_caps) {
		if (V > 1 && (small || d->cfg.Nx%(uint32_t)V != 0u)) return;
		auto push = [&](int c, int block) { if (d->ncand < 64) { d->cand[d->ncand].V = V; d->cand[d->ncand].P = P; d->cand[d->ncand].blocks_per_cu = c; d->cand[d->ncand].block = block; d->cand[d->ncand].ms[0] = d->cand[d->ncand].ms[1] = d->cand[d->ncand].ms[2] = 1e30f; d->ncand++; } };
		for (int c : caps) push(c, 0);
		// rows that do not fill whole 256-lane workgroups: smaller workgroups waste fewer lanes at the row end
		if ((d->cfg.Nx/(uint32_t)V)%256u != 0u && d->cfg.Nx/(uint32_t)V > 128u) for (int c : ragged_caps) { push(c, 128); push(c, 64); }
	};
	// P = 2 (two cells at a time in packed arithmetic): a third fewer vector instructions, 1.3-1.6x the registers, so the
	// register file already limits the waves per CU; +10 % for the arithmetic-heavy variants with 16-bit DDFs
	if (!half) { add(1, 1, {2, 3, 4, 5, 0}, {3, 0}); add(2, 1, {2, 3, 4, 0}, {2, 3}); add(4, 1, {0, 5}, {}); add(2, 2, {3, 0}, {}); }
	else { add(4, 1, {0, 4, 5, 6, 7, 8}, {0, 4}); add(2, 1, {0, 6}, {6}); add(1, 1, {0, 4, 8}, {8}); add(4, 2, {0, 4}, {0}); add(2, 2, {0}, {}); }
}

static int launch_collide_geometry(lbmhip_domain* d, const Args& base, int V, int P, int blocks_per_cu, int block_lanes, uint32_t z0, uint32_t z1, bool fields_only) {
	const uint32_t lanes_per_row = (d->cfg.Nx+(uint32_t)V-1u)/(uint32_t)V;
	uint32_t block = block_lanes ? (uint32_t)block_lanes : 256u;
	const uint32_t need = (lanes_per_row+63u)/64u*64u;
	if (block > need) block = need;
	if (blocks_per_cu > 0) blocks_per_cu = std::max(1, blocks_per_cu*256/(int)block); // the cap is quoted for 256-lane workgroups
	Args a = base;
	a.z0 = z0; a.z1 = z1;
	a.xchunks = (lanes_per_row+block-1u)/block;
	// one grid dimension carries at most 2^32-1 work-items: x-chunks and y go to grid.x, lattice planes to grid.y
	const uint64_t blocks_x = (uint64_t)a.xchunks*d->cfg.Ny;
	NEED(blocks_x*block < 0xFFFFFFFFull && (z1-z0) <= 65535u, "lattice too large for one launch (%llu x %u workgroups)", (unsigned long long)blocks_x, z1-z0);
	const CollideKernel kernel = find_collide(d, V, P, fields_only);
	NEED(kernel.fn || kernel.jit, "no compiled kernel for D?Q%u storage %u collision %u", d->cfg.velocity_set, d->cfg.storage, d->cfg.collision);
	// dynamic LDS is not used by the kernel; reserving it caps the workgroups resident per CU (160 KiB LDS per CU)
	const uint32_t lds = blocks_per_cu > 0 ? ((160u*1024u/(uint32_t)blocks_per_cu)&~255u) : 0u;
	if (kernel.jit) {
		void* params[] = {(void*)&a};
		HIP_TRY(hipModuleLaunchKernel(kernel.jit, (uint32_t)blocks_x, z1-z0, 1u, block, 1u, 1u, lds, d->stream, params, nullptr));
	} else HIP_TRY(kernel.fn(a, dim3((uint32_t)blocks_x, z1-z0), dim3(block), lds, d->stream));
	if (!fields_only) d->uses_jit = kernel.jit != nullptr;
	return 0;
}

static int launch_collide_planes(lbmhip_domain* d, uint64_t t, float fx, float fy, float fz, uint32_t z0, uint32_t z1, bool fields_only) {
	const uint32_t hz = d->cfg.Dz > 1u;
	if (z0 < hz) z0 = hz;
	if (z1 > d->cfg.Nz-hz) z1 = d->cfg.Nz-hz;
	if (z0 >= z1) return 0;
	const Args a = make_args(d, t, fx, fy, fz);
	DeviceGuard g(d->cfg.device);
	int V, P, blocks_per_cu;
	default_geometry(d, fields_only, &V, &P, &blocks_per_cu);
	const bool user_set = d->cells_per_thread != 0 || d->cells_together != 0 || d->blocks_per_cu >= 0 || d->block != 0;
	if (fields_only || user_set || !d->autotune) return launch_collide_geometry(d, a, V, P, blocks_per_cu, d->block, z0, z1, fields_only);
	if (d->tuned) return launch_collide_geometry(d, a, d->tuned_V, d->tuned_P, d->tuned_blocks_per_cu, d->tuned_block, z0, z1, false);

	// Geometry search: every full-lattice launch of the first 4*ncand steps runs one candidate between two events; all
	// candidates compute the same bits, so the search is invisible in the results.  Lattices too small to time are left alone.
	const uint64_t cells = (uint64_t)d->cfg.Nx*d->cfg.Ny*(z1-z0);
	const bool searchable = d->N >= (1ull<<21) && 4u*(uint64_t)(z1-z0) >= 3u*(uint64_t)(d->cfg.Nz-2u*hz) && cells > 0u;
	if (!searchable) return launch_collide_geometry(d, a, V, P, blocks_per_cu, 0, z0, z1, false);
	if (d->ncand == 0) build_candidates(d);
	// every candidate runs kTuneReps+1 launches in a row: the first is not timed (it starts from the memory and cache state
	// the previous candidate left behind), the others are timed one by one and their median counts
	if (d->tune_pending >= 0) { // collect the previous launch's time
		float ms = 0.0f;
		HIP_TRY(hipEventSynchronize(d->ev_tune1));
		HIP_TRY(hipEventElapsedTime(&ms, d->ev_tune0, d->ev_tune1));
		const int rep = (d->tune_launches-1)%(kTuneReps+1);
		if (rep > 0) d->cand[d->tune_pending].ms[rep-1] = ms;
		d->tune_pending = -1;
		if (d->tune_launches >= (kTuneReps+1)*d->ncand) {
			int best = 0;
			auto median = [&](int k) { const float* m = d->cand[k].ms; return std::max(std::min(m[0], m[1]), std::min(std::max(m[0], m[1]), m[2])); };
			for (int k = 1; k < d->ncand; k++) if (median(k) < median(best)) best = k;
			if (getenv("LBMHIP_TUNE_VERBOSE")) for (int k = 0; k < d->ncand; k++)
				fprintf(stderr, "lbmhip tune: cells/lane %d together %d blocks/CU<=%d block %d: %.4f %.4f %.4f ms%s\n", d->cand[k].V, d->cand[k].P, d->cand[k].blocks_per_cu,
				        d->cand[k].block, d->cand[k].ms[0], d->cand[k].ms[1], d->cand[k].ms[2], k == best ? "  <- kept" : "");
			d->tuned_V = d->cand[best].V; d->tuned_P = d->cand[best].P; d->tuned_blocks_per_cu = d->cand[best].blocks_per_cu; d->tuned_block = d->cand[best].block; d->tuned = 1;
			return launch_collide_geometry(d, a, d->tuned_V, d->tuned_P, d->tuned_blocks_per_cu, d->tuned_block, z0, z1, false);
		}
	}
	const int k = d->tune_launches/(kTuneReps+1);
	HIP_TRY(hipEventRecord(d->ev_tune0, d->stream));
	const int status = launch_collide_geometry(d, a, d->cand[k].V, d->cand[k].P, d->cand[k].blocks_per_cu, d->cand[k].block, z0, z1, false);
	if (status) return status;
	HIP_TRY(hipEventRecord(d->ev_tune1, d->stream));
	d->tune_pending = k;
	d->tune_launches++;
	return 0;
}

// the kernels' own load / store conversions (lbm.cpp:561-609, kernel.cpp:858-869) applied to an array
template <int ST> __global__ __launch_bounds__(256) void k_convert_ddfs(const void* in, void* out, const uint64_t count, const int to_storage, const int pairs) {
	using S = Store<ST>; using raw = typename S::raw;
	const uint64_t stride = (uint64_t)gridDim.x*blockDim.x, first = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
	if (!pairs) {
		for (uint64_t i = first; i < count; i += stride) {
			if (to_storage) static_cast<raw*>(out)[i] = S::encode(static_cast<const float*>(in)[i]);
			else static_cast<float*>(out)[i] = S::decode(static_cast<const raw*>(in)[i]);
		}
	} else { // two neighbouring values at a time, the packed forms used when a lane collides two cells together; count is even
		typedef typename S::raw2 raw2;
		if constexpr (ST != ST_FP32) if (pairs == 2) { // each half of a dword on its own, converted in place (one cell after the other with two cells per lane)
			for (uint64_t i = first; 2u*i < count; i += stride) {
				if (to_storage) {
					const float2v x = static_cast<const float2v*>(in)[i];
					float v[2] = {x.x, x.y};
					S::template prescale<2>(v);
					uint32_t w = 0xDEADBEEFu; // whichever half is written first has to survive the writing of the other
					if (i&1u) { S::encode_half(w, 0, v[0]); S::encode_half(w, 1, v[1]); } else { S::encode_half(w, 1, v[1]); S::encode_half(w, 0, v[0]); }
					static_cast<uint32_t*>(out)[i] = w;
				} else {
					const uint32_t w = static_cast<const uint32_t*>(in)[i];
					float2v x; x.x = S::decode_half(w, 0); x.y = S::decode_half(w, 1);
					static_cast<float2v*>(out)[i] = x;
				}
			}
			return;
		}
		for (uint64_t i = first; 2u*i < count; i += stride) {
			if (to_storage) static_cast<raw2*>(out)[i] = S::encode2(static_cast<const float2v*>(in)[i]);
			else static_cast<float2v*>(out)[i] = S::decode2(static_cast<const raw2*>(in)[i]);
		}
	}
}
#define DISPATCH_Q_ST(Q_, ST_, CALL) \
	switch (Q_) { \
		case 9:  switch (ST_) { case ST_FP32: { constexpr int Q = 9,  ST = ST_FP32; CALL; } break; case ST_FP16S: { constexpr int Q = 9,  ST = ST_FP16S; CALL; } break; default: { constexpr int Q = 9,  ST = ST_FP16C; CALL; } break; } break; \
		case 15: switch (ST_) { case ST_FP32: { constexpr int Q = 15, ST = ST_FP32; CALL; } break; case ST_FP16S: { constexpr int Q = 15, ST = ST_FP16S; CALL; } break; default: { constexpr int Q = 15, ST = ST_FP16C; CALL; } break; } break; \
		case 19: switch (ST_) { case ST_FP32: { constexpr int Q = 19, ST = ST_FP32; CALL; } break; case ST_FP16S: { constexpr int Q = 19, ST = ST_FP16S; CALL; } break; default: { constexpr int Q = 19, ST = ST_FP16C; CALL; } break; } break; \
		default: switch (ST_) { case ST_FP32: { constexpr int Q = 27, ST = ST_FP32; CALL; } break; case ST_FP16S: { constexpr int Q = 27, ST = ST_FP16S; CALL; } break; default: { constexpr int Q = 27, ST = ST_FP16C; CALL; } break; } break; \
	}

// one workgroup per lattice row chunk, rows and planes as grid.x and grid.y (at most 2^32-1 work-items per grid dimension);
// sets a.xchunks and the workgroup size for the kernels that take one cell per lane
static dim3 grid_rows(const lbmhip_domain* d, Args& a, dim3* block) {
	const uint32_t lanes = std::min(256u, (d->cfg.Nx+63u)/64u*64u);
	a.xchunks = (d->cfg.Nx+lanes-1u)/lanes;
	*block = dim3(lanes);
	return dim3(a.xchunks*d->cfg.Ny, d->cfg.Nz);
}

extern "C" {

int lbmhip_initialize(lbmhip_domain* d) {
	NEED(d, "null domain");
	DeviceGuard g(d->cfg.device);
	Args a = make_args(d, 1u, 0.0f, 0.0f, 0.0f);
	const char* const pairs_env = getenv("LBMHIP_INIT_PAIRS"); // 0: one cell per lane also with 16-bit DDFs (diagnostic)
	if (d->cfg.storage != LBMHIP_FP32 && d->cfg.Nx%2u == 0u && !(pairs_env && pairs_env[0] == '0')) { // two cells per lane: dword stores
		const uint32_t lanes_per_row = d->cfg.Nx/2u, lanes = std::min(256u, (lanes_per_row+63u)/64u*64u);
		a.xchunks = (lanes_per_row+lanes-1u)/lanes;
		const dim3 grid(a.xchunks*d->cfg.Ny, d->cfg.Nz), block(lanes);
		switch (d->cfg.velocity_set*4u+d->cfg.storage) {
			case 9u*4u+1u: hipLaunchKernelGGL((k_initialize_pairs<9, ST_FP16S>), grid, block, 0, d->stream, a); break;
			case 9u*4u+2u: hipLaunchKernelGGL((k_initialize_pairs<9, ST_FP16C>), grid, block, 0, d->stream, a); break;
			case 15u*4u+1u: hipLaunchKernelGGL((k_initialize_pairs<15, ST_FP16S>), grid, block, 0, d->stream, a); break;
			case 15u*4u+2u: hipLaunchKernelGGL((k_initialize_pairs<15, ST_FP16C>), grid, block, 0, d->stream, a); break;
			case 19u*4u+1u: hipLaunchKernelGGL((k_initialize_pairs<19, ST_FP16S>), grid, block, 0, d->stream, a); break;
			case 19u*4u+2u: hipLaunchKernelGGL((k_initialize_pairs<19, ST_FP16C>), grid, block, 0, d->stream, a); break;
			case 27u*4u+1u: hipLaunchKernelGGL((k_initialize_pairs<27, ST_FP16S>), grid, block, 0, d->stream, a); break;
			default: hipLaunchKernelGGL((k_initialize_pairs<27, ST_FP16C>), grid, block, 0, d->stream, a); break;
		}
		HIP_TRY(hipGetLastError());
		return 0;
	}
	dim3 block;
	const dim3 grid = grid_rows(d, a, &block);
	DISPATCH_Q_ST((int)d->cfg.velocity_set, (int)d->cfg.storage, hipLaunchKernelGGL((k_initialize<Q, ST>), grid, block, 0, d->stream, a));
	HIP_TRY(hipGetLastError());
	return 0;
}
int lbmhip_stream_collide(lbmhip_domain* d, uint64_t t, float fx, float fy, float fz) {
	NEED(d, "null domain");
	return launch_collide_planes(d, t, fx, fy, fz, 0u, d->cfg.Nz, false);
}
// Two steps per launch: when does it apply, in which form and with which geometry?  The lattice must be one domain that is not
// split in x or y (the whole periodic lattice, or the inner planes of a z-slab between two halo exchanges), without
// UPDATE_FIELDS, with compiled instances (collide_pairs.hip).
//   k_collide_band (fuse = 2, and what automatic mode uses): band by band on each XCD, lbm_kernels.hpp.  Measured with
//   tools/fused_sweep.py on MI355X, D3Q19 SRT without extensions, MLUPs/s against one launch per step and against
//   k_collide_pair: 512^3 FP32 46 300 / 35 700 / 41 400 (88.5 / 68.3 / 79.2 % of 8 TB/s), FP16S 79 900 / 67 500 / 74 400,
//   FP16C 73 600 / 66 700 / 70 600; 256^3 FP32 43 700 / 36 000 / 37 100; 384^3 39 100 / 34 300 / 35 100; 1024^3 42 000 /
//   34 300 / 33 600, FP16S 78 500 / 71 500 / 74 500.  The planes may be of any size: the lines between the two steps are those of
//   a band.  Best with 32 tasks per plane of a band, no residency cap and the lag that lets step t's workgroups leave before
//   step t+1 asks for them (6 planes at 32 tasks); shorter lags wait, caps cost more bandwidth than the L2 hits return.
//   The TRT + subgrid channel kernel (132 registers, 3 workgroups per CU) loses to single launches in either form (65 against
//   72 %), as do the 16-bit kernels with extensions (vector issue is what bounds them); the other FP32 sets, D3Q27 and TRT without
//   extensions gain 9 to 24 points.  Automatic mode goes by PairKernels::pays, measured per instance (collide_pairs.hip);
//   fuse = 1 or 2 pairs whatever has an instance.
//   k_collide_pair (fuse = 1): any placement of the workgroups, step t+1 reads from the Infinity Cache, so lag + 4 planes must
//   fit it with room to spare (512^3 FP32: lag 4 and no cap 79.8 %, lag 3 70 %, lag 5 77 %, lag 8 65 %).
// mode 1: k_collide_pair (step t+1 reads from the Infinity Cache, any placement of the workgroups); mode 2: k_collide_band (from the L2 of
// the XCD, band by band, + k_collide_seam)
struct PairPlan { PairKernels k; int mode; uint32_t lag, xchunks, block; int blocks_per_cu; uint32_t block_rows, R, bands, steps; };
static bool pair_plan(const lbmhip_domain* d, PairPlan* plan) {
	if (d->fuse == 0 || !d->pair_sync || (d->cfg.features&LBMHIP_UPDATE_FIELDS) || d->cfg.Dx != 1u || d->cfg.Dy != 1u) return false;
	const bool half = d->cfg.storage != LBMHIP_FP32;
	const uint32_t V = half ? 2u : 1u; // the agent-scope accesses move 4 bytes per lane: one FP32 cell or two 16-bit cells
	if (d->cfg.Nx%V != 0u) return false;
	plan->k = select_collide_pair((int)d->cfg.velocity_set, (int)d->cfg.storage, (int)d->cfg.collision, (int)d->cfg.features, (int)V, (V == 2u && d->pair_cells_together == 2) ? 2 : 1);
	if (!plan->k.pair) return false;
	const uint32_t lanes_per_row = d->cfg.Nx/V;
	plan->block = std::min(256u, (lanes_per_row+63u)/64u*64u);
	// rows whose last 256-lane chunk would be less than half full: 128-lane workgroups (1600^3 FP16S, 800 lanes per row: 85.8 % of 8 TB/s
	// against 71.6 %); where the rows fill whole 256-lane chunks those are 1-5 % faster
	if (plan->block == 256u && (lanes_per_row+255u)/256u*256u-lanes_per_row >= 128u+((lanes_per_row+127u)/128u*128u-lanes_per_row)) plan->block = 128u;
	if (d->pair_block) plan->block = std::min(plan->block, (uint32_t)d->pair_block);
	plan->xchunks = (lanes_per_row+plan->block-1u)/plan->block;
	const uint64_t per_plane = (uint64_t)plan->xchunks*d->cfg.Ny;
	if (per_plane*plan->block >= 0xFFFFFFFFull || 2u*d->cfg.Nz > 65535u) return false;
	const int per_cu_max = 8*256/(int)plan->block;
	// band by band on each XCD: blocks of ceil(Ny/8) rows, at least 4 rows each
	plan->block_rows = (d->cfg.Ny+7u)/8u;
	const bool band_possible = plan->block_rows >= 4u && d->cfg.Nz >= 8u && d->xcd_round_robin;
	if (d->fuse < 0 && !(band_possible && plan->k.pays && (uint64_t)d->cfg.Nx*d->cfg.Ny >= 256ull*256ull)) return false; // automatic mode: where it was measured to pay
	plan->mode = (d->fuse != 1 && band_possible) ? 2 : 1;
	if (plan->mode == 2) {
		// tasks per plane of a band, measured with Taylor-Green lattices (tools/fused_sweep.py): 64 for 16-bit DDFs (512^3: 64 rows of one chunk
		// 80.2 % of 8 TB/s, 32 rows 76.8 %; 1024^3: 32 rows of two chunks 76.5 %, 16 rows 74.6 %) and for FP32 rows of four chunks and more
		// (1024^3: 16 rows 85.2 %, 8 rows 80.0 %), 32 for shorter FP32 rows (512^3: 8 to 64 rows 89.7 to 88.2 %)
		const uint32_t per_plane_wanted = (half || plan->xchunks > 2u) ? 64u : 32u;
		plan->R = std::max(2u, std::min(d->band_rows > 0 ? (uint32_t)d->band_rows : std::max(2u, per_plane_wanted/plan->xchunks), plan->block_rows));
		plan->bands = (plan->block_rows+plan->R-1u)/plan->R;
		plan->blocks_per_cu = d->band_blocks_per_cu >= 0 ? d->band_blocks_per_cu : 0; // every residency cap lost at 512^3 (FP32: 88.5 % of 8 TB/s without, 86.5 % with 6, 82 % with 5 workgroups per CU)
		// the workgroups of an XCD that are resident together cover `resident` tasks of its queue; step t+1 follows far enough behind
		// step t that the workgroups it waits for have left (each plane of a band is R*xchunks tasks of each step)
		const uint32_t resident = (uint32_t)(d->compute_units/8)*(uint32_t)(plan->blocks_per_cu > 0 ? std::min(plan->blocks_per_cu*256/(int)plan->block, per_cu_max) : per_cu_max);
		const uint32_t per_band_plane = plan->R*plan->xchunks;
		plan->lag = d->band_lag ? (uint32_t)d->band_lag : std::max(2u, 2u+(resident+2u*per_band_plane-1u)/(2u*per_band_plane));
		if (d->cfg.Nz < plan->lag+4u || (uint64_t)8u*per_band_plane*plan->block >= 0xFFFFFFFFull) return false;
		// time steps per launch: every step after the first leaves one row more at either edge of a block to k_collide_seam
		plan->steps = d->band_steps ? (uint32_t)d->band_steps : 2u;
		while (plan->steps > 2u && plan->block_rows < 2u*plan->steps) plan->steps--; // the last step keeps two rows of a block at least
		return true;
	}
	plan->steps = 2u;
	plan->blocks_per_cu = d->pair_blocks_per_cu >= 0 ? d->pair_blocks_per_cu : 0;
	const uint64_t resident = (uint64_t)d->compute_units*(uint64_t)(plan->blocks_per_cu > 0 ? std::min(plan->blocks_per_cu*256/(int)plan->block, per_cu_max) : per_cu_max);
	plan->lag = d->pair_lag ? (uint32_t)d->pair_lag : std::max(3u, 2u+(uint32_t)((resident+per_plane-1u)/per_plane));
	if (d->cfg.Nz < plan->lag+4u) return false;
	return true;
}
static int pairs_steps(const lbmhip_domain* d) {
	PairPlan plan;
	if (!pair_plan(d, &plan)) return 0;
	return d->cfg.Dz == 1u || d->cfg.Nz >= plan.lag+10u; // a slab pairs its planes 2 .. Nz-3
}
static void note_pair_used(lbmhip_domain* d, const PairPlan& plan) {
	d->pair_used = plan.mode; d->pair_used_lag = (int)plan.lag; d->pair_used_blocks_per_cu = plan.blocks_per_cu; d->band_used_rows = plan.mode == 2 ? (int)plan.R : 0;
	d->band_used_steps = plan.mode == 2 ? (int)plan.steps : 0;
}
// What both forms of two steps per launch share: 64 header words (error, redo count; zeroed before every launch), behind them
// k_collide_band's stamps (one per step-t task, never cleared: the launch numbers in them rise), and the redo list, which
// holds every task of step t+1 at most.
static const uint64_t kPairHeaderWords = 64u;
static int pair_buffers(lbmhip_domain* d, uint64_t stamp_words, const PairPlan& plan) {
	const uint64_t words = kPairHeaderWords+stamp_words, redo_capacity = (uint64_t)(plan.steps-1u)*d->cfg.Nz*d->cfg.Ny*plan.xchunks;
	const bool relayout = stamp_words > 0u && (words != d->band_sync_words || d->band_launches >= (1u<<28)); // another band height lays the stamps out differently
	if (relayout || words > d->band_sync_words || redo_capacity > d->band_redo_capacity) {
		HIP_TRY(hipStreamSynchronize(d->stream)); // an earlier launch may still use the old buffers
		if (d->band_sync) HIP_TRY(hipFree(d->band_sync));
		if (d->band_redo) HIP_TRY(hipFree(d->band_redo));
		d->band_sync = nullptr; d->band_redo = nullptr; d->band_sync_words = d->band_redo_capacity = 0u;
		HIP_TRY(hipMalloc((void**)&d->band_sync, words*sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void**)&d->band_redo, redo_capacity*sizeof(uint64_t)));
		HIP_TRY(fill_words(d->band_sync, 0u, words, d->stream));
		d->band_sync_words = words; d->band_redo_capacity = redo_capacity; d->band_launches = 0u;
	}
	HIP_TRY(hipMemsetAsync(d->band_sync, 0, kPairHeaderWords*sizeof(uint32_t), d->stream));
	return 0;
}
// `steps` (2..4) time steps in one k_collide_band launch; step t+s covers the planes first[s] .. first[s]+count[s]-1 (modulo Nz).
// Behind it one k_collide_seam launch per later step, in order: the rows the XCD blocks leave at their edges (s at either edge for
// step t+s: each step needs the one before on one row more) and that step's tasks on the redo list.
static int launch_collide_band_steps(lbmhip_domain* d, const PairPlan& plan, uint64_t t, float fx, float fy, float fz, uint32_t steps, const uint32_t* first, const uint32_t* count) {
	DeviceGuard g(d->cfg.device);
	const uint32_t Nz = d->cfg.Nz, Ny = d->cfg.Ny;
	NEED(steps >= 2u && steps <= plan.steps, "%u steps in one launch", steps);
	const uint64_t stamps_per_step = (uint64_t)8u*plan.bands*Nz*plan.R*plan.xchunks;
	if (pair_buffers(d, (plan.steps-1u)*stamps_per_step, plan)) return 1;
	uint32_t plane_times = 0u;
	for (uint32_t s = 0u; s < steps; s++) plane_times = std::max(plane_times, first[s]-first[0]+count[s]+s*plan.lag);
	NEED(plane_times*steps <= 65535u && plan.bands <= 65535u, "too many planes for a launch of %u steps (%u)", steps, plane_times);
	Args a = make_args(d, t, fx, fy, fz);
	a.xchunks = plan.xchunks; a.z0 = 0u; a.z1 = Nz;
	BandArgs p;
	p.error = d->band_sync; p.redo_count = d->band_sync+1; p.stamps = d->band_sync+kPairHeaderWords; p.stamps_per_step = stamps_per_step; p.redo = d->band_redo;
	p.stamp = ++d->band_launches<<3;
	p.block_rows = plan.block_rows; p.R = plan.R; p.bands = plan.bands; p.lag = plan.lag;
	p.spin_limit = d->band_spin_limit >= 0 ? (uint32_t)d->band_spin_limit : 60000u; // about 0.2 s of polling (3.4 us of sleep between two polls)
	p.steps = steps;
	for (uint32_t s = 0u; s < 4u; s++) { p.first[s] = s < steps ? first[s] : 0u; p.count[s] = s < steps ? count[s] : 0u; }
	int blocks_per_cu = plan.blocks_per_cu;
	if (blocks_per_cu > 0) blocks_per_cu = std::max(1, blocks_per_cu*256/(int)plan.block);
	const uint32_t lds = blocks_per_cu > 0 ? ((160u*1024u/(uint32_t)blocks_per_cu)&~255u)-256u : 0u; // the kernel has a word of LDS of its own
	HIP_TRY(plan.k.band(a, p, dim3(8u*plan.R*plan.xchunks, plane_times*steps, plan.bands), dim3(plan.block), lds, d->stream));
	for (uint32_t s = 1u; s < steps; s++) {
		SeamArgs q;
		q.nrows = 0u;
		for (uint32_t b = 0u; b < 8u; b++) {
			const uint32_t bs = b*plan.block_rows, be = std::min(bs+plan.block_rows, Ny);
			if (bs >= Ny) break;
			for (uint32_t y = bs; y < be; y++) if (y < bs+s || y+s >= be) q.rows[q.nrows++] = y; // not in [bs+s, be-s)
		}
		q.first = first[s]; q.count = count[s]; q.step = s; q.redo_count = p.redo_count; q.redo = p.redo;
		Args b = make_args(d, t+s, fx, fy, fz);
		b.xchunks = plan.xchunks; b.z0 = 0u; b.z1 = Nz;
		HIP_TRY(plan.k.seam(b, q, dim3(q.nrows*plan.xchunks, count[s]+1u), dim3(plan.block), 0u, d->stream));
	}
	d->band_unchecked = 1;
	if (getenv("LBMHIP_BAND_STATS")) { // diagnostic builds with LBM_BAND_EXP&32: what the waits of this launch cost
		uint32_t h[16];
		HIP_TRY(hipStreamSynchronize(d->stream));
		HIP_TRY(hipMemcpy(h, d->band_sync, sizeof(h), hipMemcpyDeviceToHost));
		unsigned long long cycles; memcpy(&cycles, h+8, 8);
		fprintf(stderr, "lbmhip band stats: %u waiting workgroups, %.2f polls each, %.1f %% polled more than once, %.1f %% more than three times, %.0f cycles per wait\n", h[4], h[4] ? (double)h[5]/h[4] : 0.0,
		        h[4] ? 100.0*h[6]/h[4] : 0.0, h[4] ? 100.0*h[7]/h[4] : 0.0, h[4] ? (double)cycles/h[4] : 0.0);
	}
	return 0;
}
static int launch_collide_pair_steps(lbmhip_domain* d, const PairPlan& plan, uint64_t t, float fx, float fy, float fz,
                                     uint32_t first1, uint32_t count1, uint32_t first2, uint32_t count2) {
	DeviceGuard g(d->cfg.device);
	if (plan.mode == 2) { const uint32_t first[2] = {first1, first2}, count[2] = {count1, count2}; return launch_collide_band_steps(d, plan, t, fx, fy, fz, 2u, first, count); }
	if (pair_buffers(d, 0u, plan)) return 1;
	Args a = make_args(d, t, fx, fy, fz);
	a.xchunks = plan.xchunks; a.z0 = 0u; a.z1 = d->cfg.Nz;
	PairArgs p;
	p.first1 = first1; p.count1 = count1; p.first2 = first2; p.count2 = count2;
	p.plane_done = d->pair_sync; p.replica_stride = d->pair_replica_stride; p.shard_done = d->pair_sync+(uint64_t)PAIR_REPLICAS*d->pair_replica_stride;
	p.error = d->band_sync; p.redo_count = d->band_sync+1; p.redo = d->band_redo;
	p.lag = plan.lag; p.spin_limit = d->band_spin_limit >= 0 ? (uint32_t)d->band_spin_limit : 60000u; // about 0.2 s of polling (3.4 us of sleep between two polls)
	HIP_TRY(hipMemsetAsync(d->pair_sync, 0, d->pair_sync_words*sizeof(uint32_t), d->stream));
	int blocks_per_cu = plan.blocks_per_cu;
	if (blocks_per_cu > 0) blocks_per_cu = std::max(1, blocks_per_cu*256/(int)plan.block);
	const uint32_t lds = blocks_per_cu > 0 ? ((160u*1024u/(uint32_t)blocks_per_cu)&~255u)-256u : 0u; // the kernel has a word of LDS of its own
	HIP_TRY(plan.k.pair(a, p, dim3(plan.xchunks*d->cfg.Ny, count1+count2), dim3(plan.block), lds, d->stream));
	// whatever the launch gave up: k_collide_seam without rows of its own
	SeamArgs q;
	q.nrows = 0u; q.first = first2; q.count = 0u; q.step = 1u; q.redo_count = p.redo_count; q.redo = p.redo;
	Args b = make_args(d, t+1u, fx, fy, fz);
	b.xchunks = plan.xchunks; b.z0 = 0u; b.z1 = d->cfg.Nz;
	HIP_TRY(plan.k.seam(b, q, dim3(64u, 1u), dim3(plan.block), 0u, d->stream));
	d->band_unchecked = 1;
	return 0;
}
int lbmhip_stream_collide_steps(lbmhip_domain* d, uint64_t t, uint32_t steps, float fx, float fy, float fz) {
	NEED(d, "null domain");
	PairPlan plan;
	uint32_t done = 0u;
	d->pair_used = 0;
	if (steps >= 2u && d->cfg.Dz == 1u && pair_plan(d, &plan)) {
		// the whole periodic lattice: every later step starts behind the seam (k_collide_pair asks for the planes p-2..p+2, k_collide_band for p-1..p+1)
		if (plan.mode == 2) {
			const uint32_t Nz = d->cfg.Nz, first[4] = {0u, 1u, 2u, 3u}, count[4] = {Nz, Nz, Nz, Nz};
			while (steps-done >= 2u) {
				const uint32_t k = std::min(steps-done, plan.steps);
				const int status = launch_collide_band_steps(d, plan, t+done, fx, fy, fz, k, first, count);
				if (status) return status;
				done += k;
			}
		} else for (; steps-done >= 2u; done += 2u) { const int status = launch_collide_pair_steps(d, plan, t+done, fx, fy, fz, 0u, d->cfg.Nz, 2u, d->cfg.Nz); if (status) return status; }
		note_pair_used(d, plan);
	}
	for (; done < steps; done++) { const int status = launch_collide_planes(d, t+done, fx, fy, fz, 0u, d->cfg.Nz, false); if (status) return status; }
	return 0;
}
int lbmhip_stream_collide_pair_planes(lbmhip_domain* d, uint64_t t, float fx, float fy, float fz, uint32_t z_begin, uint32_t z_end) {
	NEED(d, "null domain");
	NEED(z_begin <= z_end && z_end <= d->cfg.Nz, "plane range [%u, %u) outside the lattice (Nz=%u)", z_begin, z_end, d->cfg.Nz);
	NEED(d->cfg.Dz == 1u || (z_begin >= 1u && z_end <= d->cfg.Nz-1u), "plane range [%u, %u) includes a halo plane", z_begin, z_end);
	PairPlan plan;
	d->pair_used = 0;
	if (z_end-z_begin >= 5u && pair_plan(d, &plan) && z_end-z_begin >= plan.lag+6u) {
		const int status = launch_collide_pair_steps(d, plan, t, fx, fy, fz, z_begin, z_end-z_begin, z_begin+2u, z_end-z_begin-4u);
		if (status) return status;
		note_pair_used(d, plan);
		return 0;
	}
	int status = launch_collide_planes(d, t, fx, fy, fz, z_begin, z_end, false);
	if (!status && z_end-z_begin > 4u) status = launch_collide_planes(d, t+1u, fx, fy, fz, z_begin+2u, z_end-2u, false);
	return status;
}
int lbmhip_stream_collide_planes(lbmhip_domain* d, uint64_t t, float fx, float fy, float fz, uint32_t z_begin, uint32_t z_end) {
	NEED(d, "null domain");
	NEED(z_begin <= z_end && z_end <= d->cfg.Nz, "plane range [%u, %u) outside the lattice (Nz=%u)", z_begin, z_end, d->cfg.Nz);
	return launch_collide_planes(d, t, fx, fy, fz, z_begin, z_end, false);
}
int lbmhip_update_fields(lbmhip_domain* d, uint64_t t, float fx, float fy, float fz) {
	NEED(d, "null domain");
	return launch_collide_planes(d, t, fx, fy, fz, 0u, d->cfg.Nz, true);
}
int lbmhip_update_force_field(lbmhip_domain* d, uint64_t t) {
	NEED(d, "null domain");
	NEED(d->field[LBMHIP_FIELD_F], "update_force_field needs the FORCE_FIELD extension");
	DeviceGuard g(d->cfg.device);
	Args a = make_args(d, t, 0.0f, 0.0f, 0.0f);
	dim3 block;
	const dim3 grid = grid_rows(d, a, &block);
	DISPATCH_Q_ST((int)d->cfg.velocity_set, (int)d->cfg.storage, hipLaunchKernelGGL((k_update_force_field<Q, ST>), grid, block, 0, d->stream, a));
	HIP_TRY(hipGetLastError());
	return 0;
}
int lbmhip_reset_force_field(lbmhip_domain* d) {
	NEED(d, "null domain");
	NEED(d->field[LBMHIP_FIELD_F], "reset_force_field needs the FORCE_FIELD extension");
	DeviceGuard g(d->cfg.device);
	Args a = make_args(d, 0u, 0.0f, 0.0f, 0.0f);
	dim3 block;
	const dim3 grid = grid_rows(d, a, &block);
	hipLaunchKernelGGL(k_reset_force_field, grid, block, 0, d->stream, a);
	HIP_TRY(hipGetLastError());
	return 0;
}
int lbmhip_update_moving_boundaries(lbmhip_domain* d) {
	NEED(d, "null domain");
	NEED(d->cfg.features&LBMHIP_MOVING_BOUNDARIES, "update_moving_boundaries needs the MOVING_BOUNDARIES extension");
	DeviceGuard g(d->cfg.device);
	Args a = make_args(d, 0u, 0.0f, 0.0f, 0.0f);
	dim3 block;
	const dim3 grid = grid_rows(d, a, &block);
	switch (d->cfg.velocity_set) {
		case 9: hipLaunchKernelGGL((k_update_moving_boundaries<9>), grid, block, 0, d->stream, a); break;
		case 15: hipLaunchKernelGGL((k_update_moving_boundaries<15>), grid, block, 0, d->stream, a); break;
		case 19: hipLaunchKernelGGL((k_update_moving_boundaries<19>), grid, block, 0, d->stream, a); break;
		default: hipLaunchKernelGGL((k_update_moving_boundaries<27>), grid, block, 0, d->stream, a); break;
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// ---------------------------------------------------------------------------------------------- halo
static int halo_check(const lbmhip_domain* d, int what, int axis) {
	NEED(d, "null domain");
	NEED(what >= LBMHIP_HALO_FI && what <= LBMHIP_HALO_F, "unknown halo content %d", what);
	NEED(axis >= 0 && axis < 3, "axis must be 0, 1 or 2");
	const uint32_t dd = axis == 0 ? d->cfg.Dx : axis == 1 ? d->cfg.Dy : d->cfg.Dz;
	NEED(dd > 1u, "axis %d has a single domain and therefore no halo", axis);
	NEED(what != LBMHIP_HALO_F || d->field[LBMHIP_FIELD_F], "halo exchange of F needs the FORCE_FIELD extension");
	NEED(d->halo[0] && d->halo[1] && d->halo[2] && d->halo[3], "no staging buffers");
	return 0;
}
static uint64_t face_area(const lbmhip_domain* d, int axis) {
	return axis == 0 ? (uint64_t)d->cfg.Ny*d->cfg.Nz : axis == 1 ? (uint64_t)d->cfg.Nz*d->cfg.Nx : (uint64_t)d->cfg.Nx*d->cfg.Ny;
}
static uint32_t halo_bytes_per_cell(const lbmhip_domain* d, int what) { // lbm.cpp:2184-2194
	return what == LBMHIP_HALO_FI ? q_transfers(d->cfg.velocity_set)*d->fpxx_bytes : what == LBMHIP_HALO_RHO_U_FLAGS ? 17u : what == LBMHIP_HALO_FLAGS ? 1u : 12u;
}
int lbmhip_halo_bytes(const lbmhip_domain* d, int what, int axis, uint64_t* bytes) {
	NEED(bytes, "null argument");
	if (halo_check(d, what, axis)) return 1;
	*bytes = face_area(d, axis)*halo_bytes_per_cell(d, what);
	return 0;
}
static int halo_launch(lbmhip_domain* d, int what, int axis, uint64_t t, bool insert) {
	if (halo_check(d, what, axis)) return 1;
	const uint64_t A = face_area(d, axis);
	NEED(A*halo_bytes_per_cell(d, what) <= d->halo_capacity, "staging buffers too small (%llu bytes)", (unsigned long long)d->halo_capacity);
	HaloArgs h;
	h.lat = make_args(d, t, 0.0f, 0.0f, 0.0f);
	h.buf[0] = d->halo[insert ? LBMHIP_BUF_RECV_P : LBMHIP_BUF_SEND_P];
	h.buf[1] = d->halo[insert ? LBMHIP_BUF_RECV_M : LBMHIP_BUF_SEND_M];
	h.axis = (uint32_t)axis; h.A = (uint32_t)A; h.insert = insert;
	const dim3 grid((uint32_t)((A+255u)/256u), 2u), block(256);
	DeviceGuard g(d->cfg.device);
	if (what == LBMHIP_HALO_FI) {
		DISPATCH_Q_ST((int)d->cfg.velocity_set, (int)d->cfg.storage, hipLaunchKernelGGL((k_halo_fi<Q, ST>), grid, block, 0, d->stream, h));
	} else {
		hipLaunchKernelGGL(k_halo_fields, grid, block, 0, d->stream, h, (uint32_t)what);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}
int lbmhip_halo_pack(lbmhip_domain* d, int what, int axis, uint64_t t) { return halo_launch(d, what, axis, t, false); }
int lbmhip_halo_unpack(lbmhip_domain* d, int what, int axis, uint64_t t) { return halo_launch(d, what, axis, t, true); }

int lbmhip_halo_buffer(lbmhip_domain* d, int which, void** p, uint64_t* capacity) {
	NEED(d && p, "null argument");
	NEED(which >= 0 && which < 4, "unknown staging buffer %d", which);
	NEED(d->halo[which], "this domain has no staging buffers (single domain)");
	*p = d->halo[which];
	if (capacity) *capacity = d->halo_capacity;
	return 0;
}
int lbmhip_halo_set_buffers(lbmhip_domain* d, void* send_p, void* send_m, void* recv_p, void* recv_m, uint64_t capacity) {
	NEED(d, "null domain");
	if (!send_p && !send_m && !recv_p && !recv_m) { // back to the library's own buffers
		for (int k = 0; k < 4; k++) d->halo[k] = d->halo_own[k];
		d->halo_capacity = d->halo_own_capacity;
		return 0;
	}
	NEED(send_p && send_m && recv_p && recv_m, "all four staging buffers must be given");
	uint64_t need = 0u;
	for (int ax = 0; ax < 3; ax++) {
		const uint32_t dd = ax == 0 ? d->cfg.Dx : ax == 1 ? d->cfg.Dy : d->cfg.Dz;
		if (dd > 1u) need = std::max<uint64_t>(need, face_area(d, ax)*std::max<uint32_t>(q_transfers(d->cfg.velocity_set)*d->fpxx_bytes, 17u));
	}
	NEED(capacity >= need, "staging buffers of %llu bytes are too small, %llu needed", (unsigned long long)capacity, (unsigned long long)need);
	d->halo[LBMHIP_BUF_SEND_P] = send_p; d->halo[LBMHIP_BUF_SEND_M] = send_m; d->halo[LBMHIP_BUF_RECV_P] = recv_p; d->halo[LBMHIP_BUF_RECV_M] = recv_m;
	d->halo_capacity = capacity;
	return 0;
}

// Direct device-to-device copies over xGMI need peer access switched on once per ordered device pair; without it
// hipMemcpyPeerAsync still works but stages through host memory.  Failure to enable it is therefore not an error.
static void enable_peer_access(int device, int peer) {
	static std::mutex lock; // handles are owned by one thread each, but this table is shared by all of them
	static bool tried[64][64];
	if (device == peer || device < 0 || peer < 0 || device >= 64 || peer >= 64) return;
	std::lock_guard<std::mutex> hold(lock);
	if (tried[device][peer]) return;
	tried[device][peer] = true;
	DeviceGuard g(device);
	int can = 0;
	if (g.ok && hipDeviceCanAccessPeer(&can, device, peer) == hipSuccess && can) {
		const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
		if (e != hipSuccess) (void)hipGetLastError(); // already enabled, or not possible: copies fall back to staging
	}
}

int lbmhip_halo_exchange_local(lbmhip_domain** doms, uint32_t count, int what, int axis) {
	NEED(doms && count > 0u, "null argument");
	for (uint32_t k = 0; k < count; k++) NEED(doms[k], "domain %u of %u is null", k, count);
	const lbmhip_config& c0 = doms[0]->cfg;
	NEED(count == c0.Dx*c0.Dy*c0.Dz, "expected %u domains, got %u", c0.Dx*c0.Dy*c0.Dz, count);
	for (uint32_t k = 0; k < count; k++) if (halo_check(doms[k], what, axis)) return 1;
	const uint64_t bytes = face_area(doms[0], axis)*halo_bytes_per_cell(doms[0], what);
	for (uint32_t k = 0; k < count; k++) { // one message size for all: the domains must be blocks of the same decomposition
		const lbmhip_config& c = doms[k]->cfg;
		NEED(c.Nx == c0.Nx && c.Ny == c0.Ny && c.Nz == c0.Nz && c.Dx == c0.Dx && c.Dy == c0.Dy && c.Dz == c0.Dz && c.velocity_set == c0.velocity_set && c.storage == c0.storage,
		     "domain %u differs from domain 0 in size, decomposition, velocity set or DDF storage", k);
		NEED(bytes <= doms[k]->halo_capacity, "staging buffers of domain %u too small (%llu bytes, %llu needed)", k, (unsigned long long)doms[k]->halo_capacity, (unsigned long long)bytes);
	}
	for (uint32_t k = 0; k < count; k++) { // every domain's pack is complete once its own stream reaches this event
		DeviceGuard g(doms[k]->cfg.device);
		HIP_TRY(hipEventRecord(doms[k]->ev_packed, doms[k]->stream));
	}
	const uint32_t Dx = c0.Dx, Dy = c0.Dy, Dz = c0.Dz;
	for (uint32_t k = 0; k < count; k++) {
		uint32_t pos[3] = {(k%(Dx*Dy))%Dx, (k%(Dx*Dy))/Dx, k/(Dx*Dy)};
		const uint32_t dd = axis == 0 ? Dx : axis == 1 ? Dy : Dz;
		pos[axis] = (pos[axis]+1u)%dd; // periodic wrap (lbm.cpp:2130-2132)
		lbmhip_domain* d = doms[k];
		lbmhip_domain* p = doms[pos[0]+(pos[1]+pos[2]*Dy)*Dx];
		enable_peer_access(d->cfg.device, p->cfg.device);
		enable_peer_access(p->cfg.device, d->cfg.device);
		{ // d's + face -> p's - halo, on p's stream
			DeviceGuard g(p->cfg.device);
			HIP_TRY(hipStreamWaitEvent(p->stream, d->ev_packed, 0));
			HIP_TRY(hipMemcpyPeerAsync(p->halo[LBMHIP_BUF_RECV_M], p->cfg.device, d->halo[LBMHIP_BUF_SEND_P], d->cfg.device, bytes, p->stream));
		}
		{ // p's - face -> d's + halo, on d's stream
			DeviceGuard g(d->cfg.device);
			HIP_TRY(hipStreamWaitEvent(d->stream, p->ev_packed, 0));
			HIP_TRY(hipMemcpyPeerAsync(d->halo[LBMHIP_BUF_RECV_P], d->cfg.device, p->halo[LBMHIP_BUF_SEND_M], p->cfg.device, bytes, d->stream));
		}
	}
	// a domain may refill its send buffers only after its neighbours have copied them
	for (uint32_t k = 0; k < count; k++) { DeviceGuard g(doms[k]->cfg.device); HIP_TRY(hipEventRecord(doms[k]->ev_copied, doms[k]->stream)); }
	for (uint32_t k = 0; k < count; k++) {
		DeviceGuard g(doms[k]->cfg.device);
		for (uint32_t q = 0; q < count; q++) if (q != k) HIP_TRY(hipStreamWaitEvent(doms[k]->stream, doms[q]->ev_copied, 0));
	}
	return 0;
}

int lbmhip_halo_exchange_rccl(lbmhip_domain* d, void* comm, int what, int axis, int rank_minus, int rank_plus) {
	if (halo_check(d, what, axis)) return 1;
	NEED(comm, "null communicator");
	NEED(rank_minus >= 0 && rank_plus >= 0, "neighbour ranks must not be negative");
	const RcclApi& api = rccl_api();
	NEED(api.ok, "RCCL is not loaded in this process and librccl.so could not be opened");
	const size_t bytes = (size_t)(face_area(d, axis)*halo_bytes_per_cell(d, what));
	NEED(bytes <= d->halo_capacity, "staging buffers too small (%llu bytes)", (unsigned long long)d->halo_capacity);
	DeviceGuard g(d->cfg.device);
	const int kUint8 = 1; // ncclUint8
	// with a single peer on both sides the two messages are matched in the order they are issued: + face first on both ends
	int e = api.group_start();
	if (e == 0) e = api.send(d->halo[LBMHIP_BUF_SEND_P], bytes, kUint8, rank_plus, comm, d->stream);
	if (e == 0) e = api.send(d->halo[LBMHIP_BUF_SEND_M], bytes, kUint8, rank_minus, comm, d->stream);
	if (e == 0) e = api.recv(d->halo[LBMHIP_BUF_RECV_M], bytes, kUint8, rank_minus, comm, d->stream);
	if (e == 0) e = api.recv(d->halo[LBMHIP_BUF_RECV_P], bytes, kUint8, rank_plus, comm, d->stream);
	const int e_end = api.group_end(); // always close the group
	if (e == 0) e = e_end;
	if (e != 0) return fail("RCCL send/recv failed: %s", api.error_string ? api.error_string(e) : "unknown error");
	return 0;
}

// ---------------------------------------------------------------------------------------------- mesh voxelizer
// Device buffer of a mesh of T triangles, 13*T floats: T float4 boxes (k_triangle_boxes), then the three corner arrays.
// Copies the corner arrays (host) in, works out the boxes and runs the kernel on the handle's stream; returns when the flags
// (and, for a moving mesh, u and the DDFs) are written.
static const size_t kMeshFloatsPerTriangle = 13u;
static int voxelize_with_buffer(lbmhip_domain* d, float* dev, uint32_t direction, uint64_t t, uint8_t flag, const float* p0, const float* p1, const float* p2,
                                uint32_t triangles, const float* bounding_box, const float* rotation_center, const float* linear_velocity, const float* rotational_velocity) {
	const size_t T = triangles, bytes = T*3u*sizeof(float);
	float* const corners = dev+4u*T;
	hipError_t e = hipMemcpyAsync(corners, p0, bytes, hipMemcpyHostToDevice, d->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(corners+3u*T, p1, bytes, hipMemcpyHostToDevice, d->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(corners+6u*T, p2, bytes, hipMemcpyHostToDevice, d->stream);
	if (e != hipSuccess) return fail("uploading the mesh failed: %s", hipGetErrorString(e));
	VoxelArgs v;
	v.lat = make_args(d, t, 0.0f, 0.0f, 0.0f);
	v.p0 = corners; v.p1 = corners+3u*T; v.p2 = corners+6u*T;
	const char* const cull_env = getenv("LBMHIP_VOXEL_CULL"); // 0: every ray tests every triangle, as in the reference (diagnostic, tests)
	const bool cull = !(cull_env && cull_env[0] == '0');
	v.boxes = cull ? reinterpret_cast<const float4*>(dev) : nullptr;
	v.triangles = triangles; v.direction = direction; v.flag = flag;
	v.O[0] = d->cfg.Ox; v.O[1] = d->cfg.Oy; v.O[2] = d->cfg.Oz;
	for (int k = 0; k < 3; k++) {
		v.lo[k] = bounding_box[k]; v.hi[k] = bounding_box[3+k];
		v.rc[k] = rotation_center ? rotation_center[k] : 0.0f;
		v.lin[k] = linear_velocity ? linear_velocity[k] : 0.0f;
		v.rot[k] = rotational_velocity ? rotational_velocity[k] : 0.0f;
	}
	if (cull) hipLaunchKernelGGL(k_triangle_boxes, dim3((triangles+255u)/256u), dim3(256), 0, d->stream, v, reinterpret_cast<float4*>(dev));
	const uint64_t A = face_area(d, (int)direction);
	const dim3 grid((uint32_t)((A+63u)/64u)), block(64);
	DISPATCH_Q_ST((int)d->cfg.velocity_set, (int)d->cfg.storage, hipLaunchKernelGGL((k_voxelize_mesh<Q, ST>), grid, block, 0, d->stream, v));
	e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
	if (e != hipSuccess) return fail("voxelize_mesh failed: %s", hipGetErrorString(e));
	return 0;
}
static int voxelize_check(const lbmhip_domain* d, uint32_t direction, const float* p0, const float* p1, const float* p2, uint32_t triangles, const float* bounding_box) {
	NEED(d && p0 && p1 && p2 && bounding_box, "null argument");
	NEED(direction < 3u, "direction must be 0, 1 or 2");
	NEED(triangles > 0u, "mesh without triangles");
	return 0;
}
int lbmhip_voxelize_mesh(lbmhip_domain* d, uint32_t direction, uint64_t t, uint8_t flag, const float* p0, const float* p1, const float* p2, uint32_t triangles,
                         const float* bounding_box, const float* rotation_center, const float* linear_velocity, const float* rotational_velocity) {
	if (voxelize_check(d, direction, p0, p1, p2, triangles, bounding_box)) return 1;
	DeviceGuard g(d->cfg.device);
	const size_t bytes = (size_t)triangles*kMeshFloatsPerTriangle*sizeof(float);
	if (bytes > d->mesh_capacity) { // the domain keeps its triangle buffer: a mesh that moves is voxelized again every step (lbm.cpp:1645-1670)
		HIP_TRY(hipStreamSynchronize(d->stream));
		if (d->mesh) (void)hipFree(d->mesh);
		d->mesh = nullptr; d->mesh_capacity = 0u;
		HIP_TRY(hipMalloc(&d->mesh, bytes));
		d->mesh_capacity = bytes;
	}
	return voxelize_with_buffer(d, static_cast<float*>(d->mesh), direction, t, flag, p0, p1, p2, triangles, bounding_box, rotation_center, linear_velocity, rotational_velocity);
}

// The same call for callers that must not allocate device memory (serving loops, harnesses that forbid hipMalloc): the
// triangles are staged in a fixed array in the library's device image.  Every GPU holds its own copy of the array at its own
// address; host code may only use that address, looked up once per device, never the array's name.  The call copies from the
// caller's host arrays and waits for its stream, like lbmhip_voxelize_mesh, so it is not for a stream under graph capture.
static __device__ __align__(256) float scratch[kMeshFloatsPerTriangle*LBMHIP_STATIC_MESH_TRIANGLES];
static int scratch_address(int device, float** out) { // the device must be current
	static std::mutex lock;
	static float* address[64];
	NEED(device >= 0 && device < 64, "device ordinal %d is outside the table of static buffers (0..63)", device);
	std::lock_guard<std::mutex> hold(lock);
	if (!address[device]) {
		void* p = nullptr;
		HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		NEED(p, "hipGetSymbolAddress returned a null address for the static mesh buffer");
		address[device] = static_cast<float*>(p);
	}
	*out = address[device];
	return 0;
}
int lbmhip_voxelize_mesh_static(lbmhip_domain* d, uint32_t direction, uint64_t t, uint8_t flag, const float* p0, const float* p1, const float* p2, uint32_t triangles,
                                const float* bounding_box, const float* rotation_center, const float* linear_velocity, const float* rotational_velocity) {
	if (voxelize_check(d, direction, p0, p1, p2, triangles, bounding_box)) return 1;
	NEED(triangles <= LBMHIP_STATIC_MESH_TRIANGLES, "mesh of %u triangles exceeds the static buffer of %u triangles; use lbmhip_voxelize_mesh", triangles, (uint32_t)LBMHIP_STATIC_MESH_TRIANGLES);
	DeviceGuard g(d->cfg.device);
	NEED(g.ok, "cannot select device %d", d->cfg.device);
	float* dev = nullptr;
	if (scratch_address(d->cfg.device, &dev)) return 1;
	static std::mutex in_use[64]; // one array per GPU: calls of several host threads or domains on one GPU take turns
	std::lock_guard<std::mutex> hold(in_use[d->cfg.device]);
	return voxelize_with_buffer(d, dev, direction, t, flag, p0, p1, p2, triangles, bounding_box, rotation_center, linear_velocity, rotational_velocity);
}
int lbmhip_unvoxelize_mesh(lbmhip_domain* d, uint8_t flag, const float* bounding_box) {
	NEED(d && bounding_box, "null argument");
	DeviceGuard g(d->cfg.device);
	VoxelArgs v;
	memset(&v, 0, sizeof(v));
	v.lat = make_args(d, 0u, 0.0f, 0.0f, 0.0f);
	v.flag = flag;
	v.O[0] = d->cfg.Ox; v.O[1] = d->cfg.Oy; v.O[2] = d->cfg.Oz;
	for (int k = 0; k < 3; k++) { v.lo[k] = bounding_box[k]; v.hi[k] = bounding_box[3+k]; }
	dim3 block;
	const dim3 grid = grid_rows(d, v.lat, &block);
	hipLaunchKernelGGL(k_unvoxelize_mesh, grid, block, 0, d->stream, v);
	HIP_TRY(hipGetLastError());
	return 0;
}

// ---------------------------------------------------------------------------------------------- object sums
int lbmhip_object_sum(lbmhip_domain* d, int what, uint8_t flag_marker, const float* center, const uint32_t* global_size, double* sums) {
	NEED(d && sums, "null argument");
	NEED(what >= 0 && what <= 2, "what must be 0 (centre of mass), 1 (force) or 2 (torque)");
	NEED(what == 0 || d->field[LBMHIP_FIELD_F], "force and torque sums need the FORCE_FIELD extension");
	DeviceGuard g(d->cfg.device);
	const uint32_t blocks = kObjectSumBlocks;
	float4* partial = static_cast<float4*>(d->partial);
	SumArgs s;
	s.lat = make_args(d, 0u, 0.0f, 0.0f, 0.0f);
	s.partial = partial; s.what = (uint32_t)what; s.marker = flag_marker;
	s.skip_halo = global_size ? 1u : 0u;
	s.O[0] = global_size ? d->cfg.Ox : 0; s.O[1] = global_size ? d->cfg.Oy : 0; s.O[2] = global_size ? d->cfg.Oz : 0;
	s.hx = 0.5f*(float)(global_size ? global_size[0] : d->cfg.Nx);
	s.hy = 0.5f*(float)(global_size ? global_size[1] : d->cfg.Ny);
	s.hz = 0.5f*(float)(global_size ? global_size[2] : d->cfg.Nz);
	s.cx = (what == 2 && center) ? center[0] : 0.0f;
	s.cy = (what == 2 && center) ? center[1] : 0.0f;
	s.cz = (what == 2 && center) ? center[2] : 0.0f;
	hipLaunchKernelGGL(k_object_sum, dim3(blocks), dim3(256), 0, d->stream, s);
	hipError_t e = hipGetLastError();
	std::vector<float4> host(blocks);
	if (e == hipSuccess) e = hipMemcpyAsync(host.data(), partial, blocks*sizeof(float4), hipMemcpyDeviceToHost, d->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
	if (e != hipSuccess) return fail("object sum failed: %s", hipGetErrorString(e));
	sums[0] = sums[1] = sums[2] = sums[3] = 0.0;
	for (uint32_t b = 0u; b < blocks; b++) { sums[0] += host[b].x; sums[1] += host[b].y; sums[2] += host[b].z; sums[3] += host[b].w; }
	return 0;
}

// ---------------------------------------------------------------------------------------------- DDF storage conversion
int lbmhip_convert_ddfs(int device, uint32_t storage, int to_storage, int pairs, const void* in, void* out, uint64_t count) {
	NEED(in && out, "null argument");
	NEED(storage <= LBMHIP_FP16C, "unknown DDF storage %u", storage);
	NEED(!pairs || count%2u == 0u, "conversion in pairs needs an even count");
	if (count == 0u) return 0;
	DeviceGuard g(device);
	NEED(g.ok, "cannot select device %d", device);
	const uint64_t raw_bytes = count*storage_bytes(storage), float_bytes = count*4u;
	const uint64_t in_bytes = to_storage ? float_bytes : raw_bytes, out_bytes = to_storage ? raw_bytes : float_bytes;
	void* din = nullptr; void* dout = nullptr;
	hipError_t e = hipMalloc(&din, in_bytes);
	if (e == hipSuccess) e = hipMalloc(&dout, out_bytes);
	if (e == hipSuccess) e = hipMemcpy(din, in, in_bytes, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		const dim3 grid((uint32_t)std::min<uint64_t>((count+255u)/256u, 8192u)), block(256);
		switch (storage) {
			case LBMHIP_FP32: hipLaunchKernelGGL((k_convert_ddfs<ST_FP32>), grid, block, 0, nullptr, din, dout, count, to_storage, pairs); break;
			case LBMHIP_FP16S: hipLaunchKernelGGL((k_convert_ddfs<ST_FP16S>), grid, block, 0, nullptr, din, dout, count, to_storage, pairs); break;
			default: hipLaunchKernelGGL((k_convert_ddfs<ST_FP16C>), grid, block, 0, nullptr, din, dout, count, to_storage, pairs); break;
		}
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost);
	if (din) (void)hipFree(din);
	if (dout) (void)hipFree(dout);
	if (e != hipSuccess) return fail("lbmhip_convert_ddfs failed: %s", hipGetErrorString(e));
	return 0;
}

// ---------------------------------------------------------------------------------------------- timing
int lbmhip_timer_start(lbmhip_domain* d) {
	NEED(d, "null domain");
	DeviceGuard g(d->cfg.device);
	HIP_TRY(hipEventRecord(d->ev_start, d->stream));
	return 0;
}
int lbmhip_timer_stop(lbmhip_domain* d, float* ms) {
	NEED(d && ms, "null argument");
	DeviceGuard g(d->cfg.device);
	HIP_TRY(hipEventRecord(d->ev_stop, d->stream));
	HIP_TRY(hipEventSynchronize(d->ev_stop));
	HIP_TRY(hipEventElapsedTime(ms, d->ev_start, d->ev_stop));
	return pair_check(d);
}

} // extern "C"
