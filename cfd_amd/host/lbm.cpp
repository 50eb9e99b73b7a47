// lbm.cpp - host sequencing of the reference's LBM class (FluidX3D-master/src/lbm.cpp:997-1500, 2120-2194) over the C ABI.
#include "lbm.hpp"

#include <climits>
#include <cstring>
#include <sys/stat.h>
#include <unistd.h>

Units units;
Info info;
std::atomic_bool running(true);
std::atomic_bool key_P(true);
std::atomic_bool key_9(false);
vector<string> main_arguments;

#if defined(D2Q9)
static const uint velocity_set = 9u;
#elif defined(D3Q15)
static const uint velocity_set = 15u;
#elif defined(D3Q19)
static const uint velocity_set = 19u;
#else
static const uint velocity_set = 27u;
#endif
#if defined(FP16S)
static const uint storage = LBMHIP_FP16S;
#elif defined(FP16C)
static const uint storage = LBMHIP_FP16C;
#else
static const uint storage = LBMHIP_FP32;
#endif
#if defined(TRT)
static const uint collision = LBMHIP_TRT;
#else
static const uint collision = LBMHIP_SRT;
#endif

uint lbm_features() {
	uint f = 0u;
#ifdef UPDATE_FIELDS
	f |= LBMHIP_UPDATE_FIELDS;
#endif
#ifdef VOLUME_FORCE
	f |= LBMHIP_VOLUME_FORCE;
#endif
#ifdef MOVING_BOUNDARIES
	f |= LBMHIP_MOVING_BOUNDARIES;
#endif
#ifdef EQUILIBRIUM_BOUNDARIES
	f |= LBMHIP_EQUILIBRIUM_BOUNDARIES;
#endif
#ifdef FORCE_FIELD
	f |= LBMHIP_FORCE_FIELD;
#endif
#ifdef SUBGRID
	f |= LBMHIP_SUBGRID;
#endif
	return f;
}
uint bytes_per_cell_host() { return 17u+((lbm_features()&LBMHIP_FORCE_FIELD) ? 12u : 0u); }
uint bytes_per_cell_device() { return lbmhip_bytes_per_cell_device(velocity_set, storage, lbm_features()); }
uint bandwidth_bytes_per_cell_device() { return lbmhip_bandwidth_bytes_per_cell(velocity_set, storage, lbm_features()); }
uint3 resolution(const float3 box_aspect_ratio, const uint memory) {
	const float per_cell_MB = (float)bytes_per_cell_device()/1048576.0f;
	if (velocity_set == 9u) {
		const float scaling = std::sqrt((float)memory/(box_aspect_ratio.x*box_aspect_ratio.y*per_cell_MB));
		return uint3(to_uint(scaling*box_aspect_ratio.x), to_uint(scaling*box_aspect_ratio.y), 1u);
	}
	const float scaling = std::cbrt((float)memory/(box_aspect_ratio.x*box_aspect_ratio.y*box_aspect_ratio.z*per_cell_MB));
	return uint3(to_uint(scaling*box_aspect_ratio.x), to_uint(scaling*box_aspect_ratio.y), to_uint(scaling*box_aspect_ratio.z));
}

string get_exe_path() {
	char buf[PATH_MAX];
	const ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf)-1);
	string p = n > 0 ? string(buf, (size_t)n) : string("./");
	const size_t slash = p.rfind('/');
	return slash == string::npos ? string("./") : p.substr(0, slash+1);
}
void create_folder(const string& path_to_file) {
	for (size_t i = 1; i < path_to_file.size(); i++) if (path_to_file[i] == '/') mkdir(path_to_file.substr(0, i).c_str(), 0755);
}
string default_filename(const string& path, const string& name, const string& extension, const ulong t) { // <path or exe/export/><name>-<9 digit t><ext>
	char stamp[32]; std::snprintf(stamp, sizeof(stamp), "%09llu", (unsigned long long)t);
	return (path.empty() ? get_exe_path()+"export/" : path)+create_file_extension((name.empty() ? "file" : name)+"-"+stamp, extension);
}
string default_filename(const string& name, const string& extension, const ulong t) { return default_filename("", name, extension, t); }

static void check(const int status) { if (status) print_error(lbmhip_last_error()); }

// Binary STL (utilities.hpp:4530-4572 reads the same format): 80 header bytes, the triangle count, then 50 bytes per triangle -
// normal, three corners of three floats each, two attribute bytes.  Corners are rotated as they are read; the mesh is then
// scaled (size 0: to fit the box; > 0: longest side to that length; < 0: by -size) about the middle of its bounding box
// (reposition) or about the origin, and moved to `center`.  The arithmetic per corner, center+scale*(shift+p), is the
// reference's: the voxelizer's flags depend on every bit of it.
namespace {
vector<char> whole_file(const string& filename) {
	std::ifstream in(filename, std::ios::in|std::ios::binary);
	if (in.fail()) print_error("File \""+filename+"\" does not exist!");
	return vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
float3 stl_corner(const vector<char>& bytes, const uint triangle, const int corner) {
	float c[3];
	std::memcpy(c, bytes.data()+84ull+50ull*(ulong)triangle+12ull*(ulong)(corner+1), sizeof(c));
	return float3(c[0], c[1], c[2]);
}
} // namespace
Mesh* read_stl_raw(const string& path, const bool reposition, const float3& box_size, const float3& center, const float3x3& rotation, const float size) {
	const string filename = create_file_extension(path, ".stl");
	const vector<char> bytes = whole_file(filename);
	if (bytes.size() < 84u) print_error("File \""+filename+"\" is corrupt!");
	uint count = 0u;
	std::memcpy(&count, bytes.data()+80, sizeof(count));
	if (count == 0u || bytes.size() != 84ull+50ull*(ulong)count) print_error("File \""+filename+"\" is corrupt or unsupported! Only binary .stl files are supported.");
	print_info("Loading \""+filename+"\" with "+to_string(count)+" triangles.");
	Mesh* const mesh = new Mesh(count, center);
	float3* const corners[3] = {mesh->p0.data(), mesh->p1.data(), mesh->p2.data()};
	for (int k = 0; k < 3; k++) for (uint t = 0u; t < count; t++) corners[k][t] = rotation*stl_corner(bytes, t, k);
	mesh->find_bounds();
	float scale = -size;                                                   // relative to the file's own size
	if (size == 0.0f) scale = mesh->get_scale_for_box_fit(box_size);       // as large as the box allows
	else if (size > 0.0f) scale = size/mesh->get_max_size();               // longest side of the bounding box
	const float3 shift = reposition ? -0.5f*(mesh->pmin+mesh->pmax) : float3(0.0f);
	for (int k = 0; k < 3; k++) for (uint t = 0u; t < count; t++) corners[k][t] = center+scale*(shift+corners[k][t]);
	mesh->find_bounds();
	return mesh;
}

// ---------------------------------------------------------------------------------------------- LBM_Domain
LBM_Domain::LBM_Domain(const int device, const uint Nx, const uint Ny, const uint Nz, const uint Dx, const uint Dy, const uint Dz, const int Ox, const int Oy, const int Oz, const float nu, const float fx, const float fy, const float fz)
	: Nx(Nx), Ny(Ny), Nz(Nz), Dx(Dx), Dy(Dy), Dz(Dz), Ox(Ox), Oy(Oy), Oz(Oz), nu(nu), fx(fx), fy(fy), fz(fz) {
	lbmhip_config c;
	c.Nx = Nx; c.Ny = Ny; c.Nz = Nz; c.Dx = Dx; c.Dy = Dy; c.Dz = Dz; c.Ox = Ox; c.Oy = Oy; c.Oz = Oz;
	c.velocity_set = velocity_set; c.storage = storage; c.collision = collision; c.features = lbm_features(); c.nu = nu; c.device = device;
	check(lbmhip_domain_create(&c, &handle));
	const ulong N = get_N();
	rho = Memory<float>(handle, LBMHIP_FIELD_RHO, N, 1u, 1.0f);
	u = Memory<float>(handle, LBMHIP_FIELD_U, N, 3u);
	flags = Memory<uchar>(handle, LBMHIP_FIELD_FLAGS, N, 1u);
	if (lbm_features()&LBMHIP_FORCE_FIELD) F = Memory<float>(handle, LBMHIP_FIELD_F, N, 3u);
}
LBM_Domain::~LBM_Domain() { lbmhip_domain_destroy(handle); }
uint LBM_Domain::get_velocity_set() const { return velocity_set; }
void LBM_Domain::enqueue_initialize() { check(lbmhip_initialize(handle)); }
void LBM_Domain::enqueue_stream_collide() { check(lbmhip_stream_collide(handle, t, fx, fy, fz)); }
void LBM_Domain::enqueue_stream_collide_steps(const uint steps) { check(lbmhip_stream_collide_steps(handle, t, steps, fx, fy, fz)); }
void LBM_Domain::enqueue_stream_collide_pair_planes(const uint z0, const uint z1) { check(lbmhip_stream_collide_pair_planes(handle, t, fx, fy, fz, z0, z1)); }
void LBM_Domain::enqueue_stream_collide_planes(const uint z0, const uint z1) { check(lbmhip_stream_collide_planes(handle, t, fx, fy, fz, z0, z1)); }
void LBM_Domain::enqueue_update_fields() { // lbm.cpp:238-246
	if (lbm_features()&LBMHIP_UPDATE_FIELDS) return;
	if (t != t_last_update_fields) { check(lbmhip_update_fields(handle, t, fx, fy, fz)); t_last_update_fields = t; }
}
void LBM_Domain::enqueue_update_force_field() { // lbm.cpp:258-265
	if (t != t_last_force_field) { check(lbmhip_update_force_field(handle, t)); t_last_force_field = t; }
}
void LBM_Domain::enqueue_update_moving_boundaries() { check(lbmhip_update_moving_boundaries(handle)); }
void LBM_Domain::increment_time_step(const uint steps) { t += (ulong)steps; if (lbm_features()&LBMHIP_UPDATE_FIELDS) t_last_update_fields = t; }
void LBM_Domain::reset_time_step() { t = 0ull; if (lbm_features()&LBMHIP_UPDATE_FIELDS) t_last_update_fields = t; }
void LBM_Domain::finish_queue() { check(lbmhip_sync(handle)); }

// ---------------------------------------------------------------------------------------------- Memory_Container
template <typename T> LBM::Memory_Container<T>::Memory_Container(LBM* lbm, const vector<Memory<T>*>& buffers, const string& name) : lbm(lbm), buffers(buffers), name(name) {
	N = lbm->get_N(); d = buffers[0]->dimensions();
	if (N*(ulong)d == 0ull) print_error("Memory size must be larger than 0.");
	Nx = lbm->get_Nx(); Ny = lbm->get_Ny(); Nz = lbm->get_Nz(); Dx = lbm->get_Dx(); Dy = lbm->get_Dy(); Dz = lbm->get_Dz(); D = Dx*Dy*Dz;
	NxDx = Nx/Dx; NyDy = Ny/Dy; NzDz = Nz/Dz; Hx = Dx > 1u; Hy = Dy > 1u; Hz = Dz > 1u;
	local_Nx = (ulong)(NxDx+2u*Hx); local_Ny = (ulong)(NyDy+2u*Hy); local_N = local_Nx*local_Ny*(ulong)(NzDz+2u*Hz);
	x = Pointer(this, 0u); y = Pointer(this, 1u); z = Pointer(this, 2u);
}
template <typename T> void LBM::Memory_Container<T>::read_from_device() {
	if (lbm->initialized) for (LBM_Domain* dom : lbm->lbm_domain) dom->enqueue_update_fields();
	for (Memory<T>* b : buffers) b->read_from_device();
}
template <typename T> void LBM::Memory_Container<T>::write_to_device() { for (Memory<T>* b : buffers) b->write_to_device(); }
template <typename T> string LBM::Memory_Container<T>::vtk_type() const {
	if (std::is_same<T, uchar>::value) return "unsigned_char";
	if (std::is_same<T, float>::value) return "float";
	print_error("Error in vtk_type(): Type not supported.");
}
// legacy binary VTK, STRUCTURED_POINTS, big-endian, SI-scaled (lbm.hpp:298-340); this is what visualizer.py reads
template <typename T> void LBM::Memory_Container<T>::write_vtk(const string& path, const bool convert_to_si_units) {
	float spacing = 1.0f;
	T factor = (T)1;
	if (convert_to_si_units) {
		spacing = units.si_x(1.0f);
		if (name == "rho") factor = (T)units.si_rho(1.0f);
		if (name == "u") factor = (T)units.si_u(1.0f);
		if (name == "F") factor = (T)units.si_F(1.0f);
	}
	const string filename = create_file_extension(path, ".vtk");
	const float3 origin = spacing*float3(0.5f-0.5f*(float)Nx, 0.5f-0.5f*(float)Ny, 0.5f-0.5f*(float)Nz);
	const string header = "# vtk DataFile Version 3.0\nFluidX3D "+filename.substr(filename.rfind('/')+1)+"\nBINARY\nDATASET STRUCTURED_POINTS\n"
		"DIMENSIONS "+to_string(Nx)+" "+to_string(Ny)+" "+to_string(Nz)+"\n"
		"ORIGIN "+to_string(origin.x)+" "+to_string(origin.y)+" "+to_string(origin.z)+"\n"
		"SPACING "+to_string(spacing)+" "+to_string(spacing)+" "+to_string(spacing)+"\n"
		"POINT_DATA "+to_string(N)+"\nSCALARS data "+vtk_type()+" "+to_string(d)+"\nLOOKUP_TABLE default\n";
	create_folder(filename);
	std::ofstream file(filename, std::ios::out|std::ios::binary);
	if (!file) print_error("Cannot write file \""+filename+"\".");
	file.write(header.c_str(), (std::streamsize)header.length());
	const ulong chunk = 1ull<<20; // cells per chunk: SoA -> interleaved components, byte-swapped
	vector<T> out(chunk*(ulong)d);
	for (ulong c0 = 0ull; c0 < N; c0 += chunk) {
		const ulong n = std::min(chunk, N-c0);
		parallel_for(n, [&](ulong i) { for (uint c = 0u; c < d; c++) out[i*(ulong)d+c] = reverse_bytes((T)(factor*reference(c0+i, c))); });
		file.write((const char*)out.data(), (std::streamsize)(n*(ulong)d*sizeof(T)));
	}
	file.close();
	print_info("File \""+filename+"\" saved.");
}
template <typename T> void LBM::Memory_Container<T>::write_host_to_vtk(const string& path, const bool convert_to_si_units) { write_vtk(default_filename(path, name, ".vtk", lbm->get_t()), convert_to_si_units); }
template <typename T> void LBM::Memory_Container<T>::write_device_to_vtk(const string& path, const bool convert_to_si_units) { read_from_device(); write_host_to_vtk(path, convert_to_si_units); }
template class LBM::Memory_Container<float>;
template class LBM::Memory_Container<uchar>;

// ---------------------------------------------------------------------------------------------- LBM
LBM::LBM(const uint Nx, const uint Ny, const uint Nz, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N, const float particles_rho)
	: LBM(Nx, Ny, Nz, 1u, 1u, 1u, nu, fx, fy, fz, sigma, alpha, beta, particles_N, particles_rho) {}
LBM::LBM(const uint3 N, const uint Dx, const uint Dy, const uint Dz, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N, const float particles_rho)
	: LBM(N.x, N.y, N.z, Dx, Dy, Dz, nu, fx, fy, fz, sigma, alpha, beta, particles_N, particles_rho) {}
LBM::LBM(const uint3 N, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N, const float particles_rho)
	: LBM(N.x, N.y, N.z, 1u, 1u, 1u, nu, fx, fy, fz, sigma, alpha, beta, particles_N, particles_rho) {}

LBM::LBM(const uint Nx, const uint Ny, const uint Nz, const uint Dx, const uint Dy, const uint Dz, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N, const float particles_rho) {
	(void)particles_rho;
	graphics.lbm = this;
	if (Dx*Dy*Dz == 0u) print_error("You specified 0 LBM grid domains ("+to_string(Dx)+"x"+to_string(Dy)+"x"+to_string(Dz)+"). There has to be at least 1 domain in every direction. Check your input in LBM constructor.");
	const uint NDx = (Nx/Dx)*Dx, NDy = (Ny/Dy)*Dy, NDz = (Nz/Dz)*Dz; // make the resolution divisible by the domains (lbm.cpp:1043-1054)
	if (NDx != Nx || NDy != Ny || NDz != Nz) print_warning("LBM grid ("+to_string(Nx)+"x"+to_string(Ny)+"x"+to_string(Nz)+") is not equally divisible in domains ("+to_string(Dx)+"x"+to_string(Dy)+"x"+to_string(Dz)+"). Changing resolution to ("+to_string(NDx)+"x"+to_string(NDy)+"x"+to_string(NDz)+").");
	this->Nx = NDx; this->Ny = NDy; this->Nz = NDz; this->Dx = Dx; this->Dy = Dy; this->Dz = Dz;
	const uint D = Dx*Dy*Dz, Hx = Dx > 1u, Hy = Dy > 1u, Hz = Dz > 1u;
	sanity_checks_constructor(D, nu, fx, fy, fz, sigma, alpha, beta, particles_N);
	// device selection (lbm.cpp:936-995): ordinals from the command line if there is one per domain, else the first D
	// GPUs, else every domain on GPU 0
	int available = 0;
	check(lbmhip_device_count(&available));
	vector<int> devices(D, 0);
	if ((uint)main_arguments.size() == D) for (uint d = 0u; d < D; d++) devices[d] = std::atoi(main_arguments[d].c_str());
	else {
		if (!main_arguments.empty()) print_warning("Incorrect number of devices specified. Using single fastest device for all domains.");
		else if ((uint)available >= D) for (uint d = 0u; d < D; d++) devices[d] = (int)d;
		else if (D > 1u) print_warning("Not enough devices of the same type available. Using single fastest device for all domains.");
	}
	for (uint d = 0u; d < D; d++) {
		const uint x = (d%(Dx*Dy))%Dx, y = (d%(Dx*Dy))/Dx, z = d/(Dx*Dy); // d = x+(y+z*Dy)*Dx
		lbm_domain.push_back(new LBM_Domain(devices[d], this->Nx/Dx+2u*Hx, this->Ny/Dy+2u*Hy, this->Nz/Dz+2u*Hz, Dx, Dy, Dz,
			(int)(x*this->Nx/Dx)-(int)Hx, (int)(y*this->Ny/Dy)-(int)Hy, (int)(z*this->Nz/Dz)-(int)Hz, nu, fx, fy, fz));
		handles.push_back(lbm_domain.back()->handle);
	}
	{ vector<Memory<float>*> b; for (LBM_Domain* d : lbm_domain) b.push_back(&d->rho); rho = Memory_Container<float>(this, b, "rho"); }
	{ vector<Memory<float>*> b; for (LBM_Domain* d : lbm_domain) b.push_back(&d->u); u = Memory_Container<float>(this, b, "u"); }
	{ vector<Memory<uchar>*> b; for (LBM_Domain* d : lbm_domain) b.push_back(&d->flags); flags = Memory_Container<uchar>(this, b, "flags"); }
#ifdef FORCE_FIELD
	{ vector<Memory<float>*> b; for (LBM_Domain* d : lbm_domain) b.push_back(&d->F); F = Memory_Container<float>(this, b, "F"); }
#endif
}
LBM::~LBM() {
	for (LBM_Domain* d : lbm_domain) delete d;
}

void LBM::sanity_checks_constructor(const uint D, const float nu, const float fx, const float fy, const float fz, const float sigma, const float alpha, const float beta, const uint particles_N) { // lbm.cpp:1135-1269
	if ((ulong)Nx*(ulong)Ny*(ulong)Nz == 0ull) print_error("Grid point number is 0: "+to_string(Nx)+"x"+to_string(Ny)+"x"+to_string(Nz)+" = 0.");
	int available = 0;
	check(lbmhip_device_count(&available));
	ulong memory_total = 0ull;
	check(lbmhip_device_memory(0, nullptr, &memory_total));
	const ulong memory_required = get_N()/(ulong)D*(ulong)bytes_per_cell_device();
	if (memory_required > memory_total) {
		const float factor = std::cbrt((float)memory_total/(float)memory_required);
		print_error("Grid resolution ("+to_string(Nx)+", "+to_string(Ny)+", "+to_string(Nz)+") is too large: "+to_string(D)+"x "+to_string((uint)(memory_required/1048576ull))+" MB required, "+to_string(D)+"x "+to_string((uint)(memory_total/1048576ull))+" MB available. Largest possible resolution is ("+to_string((uint)(factor*(float)Nx))+", "+to_string((uint)(factor*(float)Ny))+", "+to_string((uint)(factor*(float)Nz))+"). Restart the simulation with lower resolution or on different device(s) with more memory.");
	}
	if (nu == 0.0f) print_error("Viscosity cannot be 0. Change it in setup.cpp.");
	else if (nu < 0.0f) print_error("Viscosity cannot be negative. Remove the \"-\" in setup.cpp.");
	if (velocity_set == 9u && Nz != 1u) print_error("D2Q9 is the 2D velocity set. You have to set Nz=1u in the LBM constructor! Currently you have set Nz="+to_string(Nz)+"u.");
	if (!(lbm_features()&LBMHIP_VOLUME_FORCE) && (fx != 0.0f || fy != 0.0f || fz != 0.0f)) print_error("Volume force is set in LBM constructor in main_setup(), but VOLUME_FORCE is not enabled. Uncomment \"#define VOLUME_FORCE\" in defines.hpp.");
	if (sigma != 0.0f) print_error("Surface tension is set in LBM constructor in main_setup(), but SURFACE is not enabled.");
	if (alpha != 0.0f || beta != 0.0f) print_error("Thermal diffusion/expansion coefficients are set in LBM constructor in main_setup(), but TEMPERATURE is not enabled.");
	if (particles_N > 0u) print_error("The PARTICLES extension is disabled but the number of particles is set to "+to_string(particles_N)+">0.");
}

void LBM::sanity_checks_initialization() { // host scan of the flags (lbm.cpp:1271-1352)
	const uint threads = std::max(1u, (uint)thread::hardware_concurrency());
	vector<uchar> used(threads, 0u);
	vector<char> moving(threads, 0), equilibrium(threads, 0);
	parallel_for(get_N(), threads, [&](ulong n, uint t) {
		const uchar fl = flags[n], bo = fl&(TYPE_S|TYPE_E);
		used[t] |= fl;
		if (bo&TYPE_S) moving[t] = moving[t] || (bo == TYPE_S && (u.x[n] != 0.0f || u.y[n] != 0.0f || u.z[n] != 0.0f)) || bo == (TYPE_S|TYPE_E);
		equilibrium[t] = equilibrium[t] || bo == TYPE_E;
	});
	uchar flags_used = 0u; bool moving_used = false, equilibrium_used = false;
	for (uint t = 0u; t < threads; t++) { flags_used |= used[t]; moving_used = moving_used || moving[t]; equilibrium_used = equilibrium_used || equilibrium[t]; }
	const uint f = lbm_features();
	if (!(f&LBMHIP_MOVING_BOUNDARIES) && moving_used) print_warning("Some boundary cells have non-zero velocity, but MOVING_BOUNDARIES is not enabled. If you intend to use moving boundaries, uncomment \"#define MOVING_BOUNDARIES\" in defines.hpp.");
	if ((f&LBMHIP_MOVING_BOUNDARIES) && !moving_used) print_warning("The MOVING_BOUNDARIES extension is enabled but no moving boundary cells (TYPE_S flag and velocity unequal to zero) are placed in the simulation box.");
	if (!(f&LBMHIP_EQUILIBRIUM_BOUNDARIES) && equilibrium_used) print_error("Some cells are set as equilibrium boundaries with the TYPE_E flag, but EQUILIBRIUM_BOUNDARIES is not enabled. Uncomment \"#define EQUILIBRIUM_BOUNDARIES\" in defines.hpp.");
	if ((f&LBMHIP_EQUILIBRIUM_BOUNDARIES) && !equilibrium_used) print_warning("The EQUILIBRIUM_BOUNDARIES extension is enabled but no equilibrium boundary cells (TYPE_E flag) are placed in the simulation box.");
	if (flags_used&(TYPE_F|TYPE_I|TYPE_G)) print_error("Some cells are set as fluid/interface/gas with the TYPE_F/TYPE_I/TYPE_G flags, but SURFACE is not enabled.");
	if (flags_used&TYPE_T) print_error("Some cells are set as temperature boundary with the TYPE_T flag, but TEMPERATURE is not enabled.");
}

// LBM::communicate_field (lbm.cpp:2120-2182): x, then y, then z; pack on every domain, device-to-device copies between
// neighbours (no host staging, no host barrier), unpack on every domain
void LBM::communicate(const int what) {
	const uint D[3] = {Dx, Dy, Dz};
	for (int axis = 0; axis < 3; axis++) {
		if (D[axis] <= 1u) continue;
		for (LBM_Domain* d : lbm_domain) check(lbmhip_halo_pack(d->handle, what, axis, d->get_t()));
		check(lbmhip_halo_exchange_local(handles.data(), (uint)handles.size(), what, axis));
		for (LBM_Domain* d : lbm_domain) check(lbmhip_halo_unpack(d->handle, what, axis, d->get_t()));
	}
}
// z-slabs: the DDF faces that do_two_time_steps packed at time step packed_t and left for later
void LBM::finish_packed_exchange() {
	if (!packed_pending) return;
	packed_pending = false;
	check(lbmhip_halo_exchange_local(handles.data(), (uint)handles.size(), LBMHIP_HALO_FI, 2));
	for (LBM_Domain* d : lbm_domain) check(lbmhip_halo_unpack(d->handle, LBMHIP_HALO_FI, 2, packed_t));
}
void LBM::flush_packed_exchange() {
	if (!packed_pending) return;
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_begin(d->handle));
	finish_packed_exchange();
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_end(d->handle));
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_join(d->handle));
}

void LBM::initialize() { // lbm.cpp:1354-1409
#ifndef BENCHMARK
	sanity_checks_initialization();
#endif
	for (LBM_Domain* d : lbm_domain) d->rho.write_to_device();
	for (LBM_Domain* d : lbm_domain) d->u.write_to_device();
	for (LBM_Domain* d : lbm_domain) d->flags.write_to_device();
	if (lbm_features()&LBMHIP_FORCE_FIELD) {
		for (LBM_Domain* d : lbm_domain) d->F.write_to_device();
		communicate(LBMHIP_HALO_F);
	}
	for (LBM_Domain* d : lbm_domain) d->increment_time_step(); // the halo exchanges at initialization need an odd time step
	communicate(LBMHIP_HALO_RHO_U_FLAGS);
	for (LBM_Domain* d : lbm_domain) d->enqueue_initialize();
	communicate(LBMHIP_HALO_RHO_U_FLAGS);
	communicate(LBMHIP_HALO_FI);
	for (LBM_Domain* d : lbm_domain) d->finish_queue();
	for (LBM_Domain* d : lbm_domain) d->reset_time_step();
	initialized = true;
}

void LBM::do_time_step() { // lbm.cpp:1411-1459, without the per-step host synchronisation
	const bool slabs = Dx == 1u && Dy == 1u && Dz > 1u && lbm_domain[0]->get_Nz() >= 5u;
	if (slabs) { // planes next to the halos first; their exchange runs on the communication streams beside the rest
		const uint Lz = lbm_domain[0]->get_Nz();
		for (LBM_Domain* d : lbm_domain) { d->enqueue_stream_collide_planes(1u, 2u); d->enqueue_stream_collide_planes(Lz-2u, Lz-1u); }
		for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_begin(d->handle));
		communicate(LBMHIP_HALO_FI);
		for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_end(d->handle));
		for (LBM_Domain* d : lbm_domain) d->enqueue_stream_collide_planes(2u, Lz-2u);
		for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_join(d->handle));
	} else {
		for (LBM_Domain* d : lbm_domain) d->enqueue_stream_collide();
		communicate(LBMHIP_HALO_FI);
	}
	for (LBM_Domain* d : lbm_domain) d->increment_time_step();
}

// z-slabs whose library runs the inner planes of two steps in one launch
bool LBM::pairs_slab_steps() const {
	if (!(Dx == 1u && Dy == 1u && Dz > 1u && lbm_domain[0]->get_Nz() >= 12u)) return false;
	for (const LBM_Domain* d : lbm_domain) { int v = 0; check(lbmhip_get_tuning(d->handle, "pairs_steps", &v)); if (!v) return false; }
	return true;
}
// Two steps of z-slabs: step t next to the halos and its exchange beside ONE launch per domain that runs step t on the
// inner planes and step t+1 two planes further in (lbmhip_stream_collide_pair_planes), then step t+1 on the six planes near
// the halos and its exchange.  A cell of step t+1 in plane p depends on step t in the planes p-2 .. p+2 only: same bits as
// two calls of do_time_step.
void LBM::do_two_time_steps() {
	const uint Lz = lbm_domain[0]->get_Nz();
	// communication streams: the exchange the call before left packed, then step t next to the halos and its exchange
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_begin(d->handle));
	finish_packed_exchange();
	for (LBM_Domain* d : lbm_domain) { d->enqueue_stream_collide_planes(1u, 2u); d->enqueue_stream_collide_planes(Lz-2u, Lz-1u); }
	communicate(LBMHIP_HALO_FI);
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_end(d->handle));
	// compute streams, beside them: step t on 2 .. Lz-3 and step t+1 on 4 .. Lz-5 in one launch
	for (LBM_Domain* d : lbm_domain) d->enqueue_stream_collide_pair_planes(2u, Lz-2u);
	for (LBM_Domain* d : lbm_domain) check(lbmhip_comm_join(d->handle));
	for (LBM_Domain* d : lbm_domain) d->increment_time_step();
	for (LBM_Domain* d : lbm_domain) { d->enqueue_stream_collide_planes(1u, 4u); d->enqueue_stream_collide_planes(Lz-4u, Lz-1u); }
	// the faces of step t+1 are packed here; their exchange runs beside the next pair's inner launch (or at the end of run())
	for (LBM_Domain* d : lbm_domain) check(lbmhip_halo_pack(d->handle, LBMHIP_HALO_FI, 2, d->get_t()));
	packed_t = lbm_domain[0]->get_t(); packed_pending = true;
	for (LBM_Domain* d : lbm_domain) d->increment_time_step();
}

std::atomic<unsigned long> lbm_runs_after_stop(0ul); // LBM::run calls made although `running` is false (main.cpp's watchdog reads it)
void LBM::run(const ulong steps, const ulong total_steps) { // lbm.cpp:1461-1489
	if (!running && steps > 0ull) lbm_runs_after_stop++;
	info.append(steps, total_steps, get_t());
	if (!initialized) initialize();
	Clock clock;
	const ulong batch = 64ull; // the host only waits for the GPU every `batch` steps; the reference waits after every step
	for (ulong i = 0ull; i < steps; ) {
		if (!running) break;
		const ulong n = std::min(batch, steps-i);
		clock.start();
		if (get_D() == 1u) { // no exchange between the steps: the library may run them two per launch (lbmhip_stream_collide_steps)
			lbm_domain[0]->enqueue_stream_collide_steps((uint)n);
			lbm_domain[0]->increment_time_step((uint)n);
		} else if (n >= 2ull && pairs_slab_steps()) {
			ulong k = 0ull;
			for (; k+1ull < n; k += 2ull) do_two_time_steps();
			flush_packed_exchange();
			for (; k < n; k++) do_time_step();
		} else for (ulong k = 0ull; k < n; k++) do_time_step();
		for (LBM_Domain* d : lbm_domain) d->finish_queue();
		const double dt = clock.stop()/(double)n;
		for (ulong k = 0ull; k < n; k++) info.update(dt);
		i += n;
	}
}
void LBM::update_fields() { // lbm.cpp:1491-1496
	for (LBM_Domain* d : lbm_domain) d->enqueue_update_fields();
	for (LBM_Domain* d : lbm_domain) d->finish_queue();
}
void LBM::reset() { initialized = false; }

#ifdef FORCE_FIELD
void LBM::update_force_field() { // lbm.cpp:1503-1508
	for (LBM_Domain* d : lbm_domain) d->enqueue_update_force_field();
	for (LBM_Domain* d : lbm_domain) d->finish_queue();
}
// Sums over all cells whose flags equal flag_marker (lbm.cpp:1509-1554): reduced on every domain's GPU, added on the host.
static void object_sums(LBM* lbm, const int what, const uchar flag_marker, const float3& center, double* total) {
	const uint size[3] = {lbm->get_Nx(), lbm->get_Ny(), lbm->get_Nz()};
	const float c[3] = {center.x, center.y, center.z};
	total[0] = total[1] = total[2] = total[3] = 0.0;
	for (LBM_Domain* d : lbm->lbm_domain) {
		double s[4];
		check(lbmhip_object_sum(d->handle, what, flag_marker, c, size, s));
		for (int k = 0; k < 4; k++) total[k] += s[k];
	}
}
float3 LBM::object_center_of_mass(const uchar flag_marker) {
	if (!initialized) flags.write_to_device();
	double s[4]; object_sums(this, 0, flag_marker, float3(0.0f), s);
	return s[3] > 0.0 ? float3((float)(s[0]/s[3]), (float)(s[1]/s[3]), (float)(s[2]/s[3])) : float3(0.0f);
}
float3 LBM::object_force(const uchar flag_marker) {
	update_force_field();
	double s[4]; object_sums(this, 1, flag_marker, float3(0.0f), s);
	return float3((float)s[0], (float)s[1], (float)s[2]);
}
float3 LBM::object_torque(const float3& rotation_center, const uchar flag_marker) {
	update_force_field();
	double s[4]; object_sums(this, 2, flag_marker, rotation_center, s);
	return float3((float)s[0], (float)s[1], (float)s[2]);
}
#endif
#ifdef MOVING_BOUNDARIES
void LBM::update_moving_boundaries() {
	communicate(LBMHIP_HALO_RHO_U_FLAGS);
	for (LBM_Domain* d : lbm_domain) d->enqueue_update_moving_boundaries();
	communicate(LBMHIP_HALO_FLAGS);
	for (LBM_Domain* d : lbm_domain) d->finish_queue();
}
#endif

// LBM_Domain::voxelize_mesh_on_device (lbm.cpp:344-412) for every domain, then LBM::voxelize_mesh_on_device (lbm.cpp:1645-1670)
void LBM::voxelize_mesh_on_device(const Mesh* mesh, const uchar flag, const float3& rotation_center, const float3& linear_velocity, const float3& rotational_velocity) {
	const float box[6] = {mesh->pmin.x-2.0f, mesh->pmin.y-2.0f, mesh->pmin.z-2.0f, mesh->pmax.x+2.0f, mesh->pmax.y+2.0f, mesh->pmax.z+2.0f};
	uint direction = 0u;
	if (length(rotational_velocity) == 0.0f) { // rays along the axis with the smallest bounding-box cross-section
		const float v[3] = {(box[4]-box[1])*(box[5]-box[2]), (box[5]-box[2])*(box[3]-box[0]), (box[3]-box[0])*(box[4]-box[1])};
		for (uint i = 1u; i < 3u; i++) if (v[i] < v[direction]) direction = i;
	} else { // rays along the axis closest to the rotation axis
		const float v[3] = {std::fabs(rotational_velocity.x), std::fabs(rotational_velocity.y), std::fabs(rotational_velocity.z)};
		for (uint i = 1u; i < 3u; i++) if (v[i] > v[direction]) direction = i;
	}
	vector<float> p0(3u*mesh->triangle_number), p1(3u*mesh->triangle_number), p2(3u*mesh->triangle_number);
	for (uint i = 0u; i < mesh->triangle_number; i++) {
		p0[3u*i] = mesh->p0[i].x; p0[3u*i+1u] = mesh->p0[i].y; p0[3u*i+2u] = mesh->p0[i].z;
		p1[3u*i] = mesh->p1[i].x; p1[3u*i+1u] = mesh->p1[i].y; p1[3u*i+2u] = mesh->p1[i].z;
		p2[3u*i] = mesh->p2[i].x; p2[3u*i+1u] = mesh->p2[i].y; p2[3u*i+2u] = mesh->p2[i].z;
	}
	const float rc[3] = {rotation_center.x, rotation_center.y, rotation_center.z}, lin[3] = {linear_velocity.x, linear_velocity.y, linear_velocity.z}, rot[3] = {rotational_velocity.x, rotational_velocity.y, rotational_velocity.z};
	for (LBM_Domain* d : lbm_domain) check(lbmhip_voxelize_mesh(d->handle, direction, d->get_t()+1ull, flag, p0.data(), p1.data(), p2.data(), mesh->triangle_number, box, rc, lin, rot));
#ifdef MOVING_BOUNDARIES
	if ((flag&(TYPE_S|TYPE_E)) == TYPE_S && (length(linear_velocity) > 0.0f || length(rotational_velocity) > 0.0f)) update_moving_boundaries();
#endif
	if (!initialized) { // the host copies follow the device until the first run()
		for (LBM_Domain* d : lbm_domain) { d->flags.read_from_device(); d->u.read_from_device(); }
	}
}
void LBM::unvoxelize_mesh_on_device(const Mesh* mesh, const uchar flag) {
	const float box[6] = {mesh->pmin.x, mesh->pmin.y, mesh->pmin.z, mesh->pmax.x, mesh->pmax.y, mesh->pmax.z};
	for (LBM_Domain* d : lbm_domain) check(lbmhip_unvoxelize_mesh(d->handle, flag, box));
	for (LBM_Domain* d : lbm_domain) d->finish_queue();
}
void LBM::voxelize_stl(const string& path, const float3& center, const float3x3& rotation, const float size, const uchar flag) {
	const Mesh* mesh = read_stl(path, this->size(), center, rotation, size);
	flags.write_to_device();
	voxelize_mesh_on_device(mesh, flag);
	delete mesh;
	for (LBM_Domain* d : lbm_domain) d->flags.read_from_device();
}
void LBM::voxelize_stl(const string& path, const float3x3& rotation, const float size, const uchar flag) { voxelize_stl(path, center(), rotation, size, flag); }
void LBM::voxelize_stl(const string& path, const float3& center, const float size, const uchar flag) { voxelize_stl(path, center, float3x3(1.0f), size, flag); }
void LBM::voxelize_stl(const string& path, const float size, const uchar flag) { voxelize_stl(path, center(), float3x3(1.0f), size, flag); }

// Triangle soup as legacy VTK POLYDATA, binary big-endian, positions relative to the box centre (lbm.cpp:1682-1724)
void LBM::write_mesh_to_vtk(const Mesh* mesh, const string& path, const bool convert_to_si_units) const {
	const string filename = default_filename(path, "mesh", ".vtk", get_t());
	const uint T = mesh->triangle_number;
	const float spacing = convert_to_si_units ? units.si_x(1.0f) : 1.0f;
	const float3 offset = center();
	vector<float> points(9ull*T);
	vector<int> polygons(4ull*T);
	parallel_for((ulong)T, [&](ulong i) {
		const float3 v[3] = {mesh->p0[i], mesh->p1[i], mesh->p2[i]};
		for (uint k = 0u; k < 3u; k++) {
			points[9ull*i+3ull*k] = reverse_bytes(spacing*(v[k].x-offset.x));
			points[9ull*i+3ull*k+1ull] = reverse_bytes(spacing*(v[k].y-offset.y));
			points[9ull*i+3ull*k+2ull] = reverse_bytes(spacing*(v[k].z-offset.z));
			polygons[4ull*i+1ull+k] = reverse_bytes((int)(3ull*i+k));
		}
		polygons[4ull*i] = reverse_bytes(3);
	});
	create_folder(filename);
	std::ofstream file(filename, std::ios::out|std::ios::binary);
	if (!file) print_error("Cannot write file \""+filename+"\".");
	const string head = "# vtk DataFile Version 3.0\nFluidX3D "+filename.substr(filename.rfind('/')+1)+"\nBINARY\nDATASET POLYDATA\nPOINTS "+to_string(3u*T)+" float\n";
	const string mid = "POLYGONS "+to_string(T)+" "+to_string(4u*T)+"\n";
	file.write(head.c_str(), (std::streamsize)head.length());
	file.write((const char*)points.data(), (std::streamsize)(points.size()*sizeof(float)));
	file.write(mid.c_str(), (std::streamsize)mid.length());
	file.write((const char*)polygons.data(), (std::streamsize)(polygons.size()*sizeof(int)));
	file.close();
	print_info("File \""+filename+"\" saved.");
}
void LBM::write_status(const string& path) { // lbm.cpp:1599-1643: the same "key = value" lines
	const string filename = default_filename(path, "status", ".txt", get_t());
	create_folder(filename);
	std::ofstream f(filename);
	if (!f) { print_warning("Cannot write \""+filename+"\"."); return; }
	const double total = info.runtime_total();
	f << "Grid Resolution = " << Nx << " x " << Ny << " x " << Nz << " = " << get_N() << "\n";
	f << "Grid Domains = " << Dx << " x " << Dy << " x " << Dz << " = " << get_D() << "\n";
	f << "LBM Type = D" << (velocity_set == 9u ? 2 : 3) << "Q" << velocity_set << " " << (collision == LBMHIP_TRT ? "TRT" : "SRT") << " (FP32/" << (storage == LBMHIP_FP32 ? "FP32" : storage == LBMHIP_FP16S ? "FP16S" : "FP16C") << ")\n";
	f << "Memory Usage = CPU " << (uint)(get_N()*(ulong)bytes_per_cell_host()/1048576ull) << " MB, GPU " << get_D() << "x " << (uint)(lbm_domain[0]->get_N()*(ulong)bytes_per_cell_device()/1048576ull) << " MB\n";
	f << "Maximum Allocation Size = " << (uint)(get_N()/(ulong)get_D()*(ulong)(velocity_set*(storage == LBMHIP_FP32 ? 4u : 2u))/1048576ull) << " MB\n";
	f << "Time Steps = " << get_t() << " / " << (info.steps == max_ulong ? string("infinite") : to_string(info.steps)) << "\n";
	f << "Runtime = " << print_time(total) << " (total) = " << print_time(info.runtime_lbm) << " (LBM) + " << print_time(std::max(total-info.runtime_lbm, 0.0)) << " (rendering and data evaluation)\n";
	f << "Average MLUPs/s = " << (info.runtime_lbm > 0.0 ? to_uint(1E-6*(double)get_N()*(double)get_t()/info.runtime_lbm) : 0u) << "\n";
	f << "Kinematic Viscosity = " << to_string(get_nu()) << "\n";
	f << "Relaxation Time = " << to_string(get_tau()) << "\n";
	f << "Maximum Reynolds Number = " << to_string(get_Re_max()) << "\n";
#ifdef VOLUME_FORCE
	f << "Volume Force = (" << to_string(get_fx()) << ", " << to_string(get_fy()) << ", " << to_string(get_fz()) << ")\n";
#endif
	f.close();
	print_info("File \""+filename+"\" saved.");
}
