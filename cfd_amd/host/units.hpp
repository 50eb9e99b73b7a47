// units.hpp - SI <-> lattice unit conversion.  The method names, argument orders and meanings MIRROR the reference's Units
// class (FluidX3D-master/src/units.hpp:5-161) on purpose: setups written for the reference call them and must compile
// unchanged.  The whole set is provided: base units, SI <-> lattice conversions, the dimensionless numbers and the
// *_from_* relations (each once, in `Relations`, for both unit systems: a dimensionless number or a relation between
// quantities reads the same in lattice and in SI units), Poiseuille and Stokes flow.
// Outputs are converted by their dimension: one lattice unit of a quantity with dimension m^L kg^M s^T K^H is
// dim(L, M, T, H) SI units and "si_q(v)" multiplies by it.
#pragma once
#include "utilities.hpp"

// Definitions shared by the lattice-unit members (Re, We, u_from_Re, ...) and the SI members (si_Re, si_We, si_u_from_si_Re, ...)
// of Units.  x length, u velocity, nu / mu kinematic / dynamic viscosity, rho density, sigma surface tension, g acceleration,
// f force per volume.
struct Relations {
	static float reynolds(const float x, const float u, const float nu) { return x*u/nu; }
	static float reynolds(const float x, const float u, const float mu, const float rho) { return x*u*rho/mu; }
	static float weber(const float x, const float u, const float rho, const float sigma) { return x*sq(u)*rho/sigma; }                  // inertia against surface tension
	static float froude(const float x, const float u, const float g) { return u/sqrt(x*g); }                                              // inertia against gravity
	static float capillary(const float u, const float mu, const float sigma) { return u*mu/sigma; }                                        // viscous force against surface tension
	static float bond(const float x, const float rho, const float g, const float sigma) { return sq(x)*rho*g/sigma; }                     // gravity against surface tension
	static float morton(const float rho, const float delta_rho, const float g, const float sigma, const float mu) { return g*delta_rho*sq(sq(mu))/(cb(sigma)*sq(rho)); }
	static float galilei(const float x, const float nu, const float g) { return cb(x)*g/sq(nu); }                                          // gravity against viscous force
	static float velocity_for_reynolds(const float Re, const float x, const float nu) { return Re*nu/x; }
	static float velocity_for_weber(const float We, const float x, const float sigma, const float rho) { return sqrt(We*sigma/(x*rho)); }
	static float velocity_for_froude(const float Fr, const float x, const float g) { return Fr*sqrt(x*g); }
	static float poiseuille_centre_velocity_2D(const float Q, const float R) { return 0.75f*Q/R; }          // flow rate Q per unit depth between plates 2R apart
	static float poiseuille_centre_velocity_3D(const float Q, const float R) { return 2.0f/pif*Q/sq(R); }  // flow rate Q through a pipe of radius R
};

class Units {
	float base[4] = {1.0f, 1.0f, 1.0f, 1.0f}; // one lattice unit of length, mass, time, temperature in m, kg, s, K
	static float ipow(const float b, int e) { float r = 1.0f; for (; e > 0; e--) r *= b; for (; e < 0; e++) r /= b; return r; }
	float dim(const int L, const int M, const int T, const int H = 0) const { return ipow(base[0], L)*ipow(base[1], M)*ipow(base[2], T)*ipow(base[3], H); }
	float um() const { return base[0]; }
	float ukg() const { return base[1]; }
	float us() const { return base[2]; }
	float uK() const { return base[3]; }
	void report() const { print_info("Unit Conversion: 1 cell = "+to_string(1000.0f*si_x(1.0f), 3u)+" mm, 1 s = "+to_string(t(1.0f))+" time steps"); }
public:
	// fix the base units from one length, one velocity and one density (and one temperature) given in both unit systems
	void set_m_kg_s(const float x, const float u, const float rho, const float si_x, const float si_u, const float si_rho) {
		base[0] = si_x/x;
		base[1] = si_rho/rho*cb(base[0]);
		base[2] = u/si_u*base[0];
		report();
	}
	void set_m_kg_s(const float m, const float kg, const float s) { base[0] = m; base[1] = kg; base[2] = s; report(); }
	void set_m_kg_s_K(const float x, const float u, const float rho, const float T, const float si_x, const float si_u, const float si_rho, const float si_T) {
		base[3] = si_T/T;
		set_m_kg_s(x, u, rho, si_x, si_u, si_rho);
	}
	void set_m_kg_s_K(const float m, const float kg, const float s, const float K) { base[3] = K; set_m_kg_s(m, kg, s); }

	// ---- SI -> lattice.  These feed simulation parameters, so each is evaluated in the reference's operation order
	// (units.hpp:38-58) and gives the same float; um, ukg, us, uK are the four base units.
	float x(const float si_x) const { return si_x/um(); }                                  // length
	float m(const float si_m) const { return si_m/ukg(); }                                 // mass
	ulong t(const float si_t) const { return to_ulong(si_t/us()); }                        // time, in steps
	float frequency(const float si_frequency) const { return si_frequency*us(); }
	float omega(const float si_omega) const { return si_omega*us(); }                      // angular frequency
	float u(const float si_u) const { return si_u*us()/um(); }                             // velocity
	float rho(const float si_rho) const { return si_rho*cb(um())/ukg(); }                  // density
	float Q(const float si_Q) const { return si_Q*us()/cb(um()); }                         // flow rate
	float nu(const float si_nu) const { return si_nu*us()/sq(um()); }                      // kinematic viscosity
	float mu(const float si_mu) const { return si_mu*us()*um()/ukg(); }                    // dynamic viscosity
	float g(const float si_g) const { return si_g/um()*sq(us()); }                         // acceleration
	float f(const float si_f) const { return si_f*sq(um()*us())/ukg(); }                   // force per volume
	float f(const float si_rho, const float si_g) const { return si_rho*si_g*sq(um()*us())/ukg(); } // ... given as density times acceleration
	float F(const float si_F) const { return si_F*sq(us())/(ukg()*um()); }                 // force
	float M(const float si_M) const { return si_M*sq(us())/(ukg()*sq(um())); }             // torque
	float sigma(const float si_sigma) const { return si_sigma*sq(us())/ukg(); }            // surface tension
	float T(const float si_T) const { return si_T/uK(); }                                  // temperature
	float alpha(const float si_alpha) const { return si_alpha*us()/sq(um()); }             // thermal diffusion coefficient
	float beta(const float si_beta) const { return si_beta*uK(); }                         // thermal expansion coefficient

	// ---- lattice -> SI
	float si_x(const uint x) const { return (float)x*dim(1, 0, 0); }
	float si_x(const float x) const { return x*dim(1, 0, 0); }
	float si_m(const float m) const { return m*dim(0, 1, 0); }
	float si_t(const ulong t) const { return (float)t*dim(0, 0, 1); }
	float si_frequency(const float frequency) const { return frequency*dim(0, 0, -1); }
	float si_V(const float V) const { return V*dim(3, 0, 0); }
	float si_u(const float u) const { return u*dim(1, 0, -1); }
	float si_rho(const float rho) const { return rho*dim(-3, 1, 0); }
	float si_p(const float p) const { return p*dim(-1, 1, -2); }
	float si_Q(const float Q) const { return Q*dim(3, 0, -1); }
	float si_nu(const float nu) const { return nu*dim(2, 0, -1); }
	float si_g(const float g) const { return g*dim(1, 0, -2); }
	float si_f(const float f) const { return f*dim(-2, 1, -2); }
	float si_F(const float F) const { return F*dim(1, 1, -2); }
	float si_M(const float M) const { return M*dim(2, 1, -2); }
	float si_sigma(const float sigma) const { return sigma*dim(0, 1, -2); }
	float si_T(const float T) const { return T*dim(0, 0, 0, 1); }
	float si_alpha(const float alpha) const { return alpha*dim(2, 0, -1); }
	float si_beta(const float beta) const { return beta*dim(0, 0, 0, -1); }

	// ---- dimensionless numbers, from lattice quantities and from SI quantities (units.hpp:79-90, 137-147)
	float Re(const float si_Re) const { return si_Re; }
	float Re(const float x, const float u, const float nu) const { return Relations::reynolds(x, u, nu); }
	float Re(const float x, const float u, const float mu, const float rho) const { return Relations::reynolds(x, u, mu, rho); }
	float We(const float x, const float u, const float rho, const float sigma) const { return Relations::weber(x, u, rho, sigma); }
	float Fr(const float x, const float u, const float g) const { return Relations::froude(x, u, g); }
	float Ca(const float u, const float mu, const float sigma) const { return Relations::capillary(u, mu, sigma); }
	float Ca(const float u, const float rho, const float nu, const float sigma) const { return Relations::capillary(u, rho*nu, sigma); }
	float Bo(const float x, const float rho, const float g, const float sigma) const { return Relations::bond(x, rho, g, sigma); }
	float Mo(const float rho, const float delta_rho, const float g, const float sigma, const float mu) const { return Relations::morton(rho, delta_rho, g, sigma, mu); }
	float Ga(const float x, const float nu, const float g) const { return Relations::galilei(x, nu, g); }
	float Ga(const float x, const float rho, const float nu, const float f) const { return cb(x)*f/(sq(nu)*rho); } // with g = f/rho
	float Ma(const float u) const { return u/0.57735027f; } // against the lattice speed of sound 1/sqrt(3)
	float si_Re(const float Re) const { return Re; }
	float si_Re(const float si_x, const float si_u, const float si_nu) const { return Relations::reynolds(si_x, si_u, si_nu); }
	float si_Re(const float si_x, const float si_u, const float si_mu, const float si_rho) const { return Relations::reynolds(si_x, si_u, si_mu, si_rho); }
	float si_We(const float si_x, const float si_u, const float si_rho, const float si_sigma) const { return Relations::weber(si_x, si_u, si_rho, si_sigma); }
	float si_Fr(const float si_x, const float si_u, const float si_g) const { return Relations::froude(si_x, si_u, si_g); }
	float si_Ca(const float si_u, const float si_mu, const float si_sigma) const { return Relations::capillary(si_u, si_mu, si_sigma); }
	float si_Ca(const float si_u, const float si_rho, const float si_nu, const float si_sigma) const { return Relations::capillary(si_u, si_rho*si_nu, si_sigma); }
	float si_Bo(const float si_x, const float si_rho, const float si_g, const float si_sigma) const { return Relations::bond(si_x, si_rho, si_g, si_sigma); }
	float si_Mo(const float si_rho, const float si_delta_rho, const float si_g, const float si_sigma, const float si_mu) const { return Relations::morton(si_rho, si_delta_rho, si_g, si_sigma, si_mu); }
	float si_Ga(const float si_x, const float si_nu, const float si_g) const { return Relations::galilei(si_x, si_nu, si_g); }
	float si_Ga(const float si_x, const float si_rho, const float si_nu, const float si_f) const { return cb(si_x)*si_f/(sq(si_nu)*si_rho); }

	// ---- relations in lattice units (units.hpp:91-112): equation of state p = (rho-1)/3 (speed of sound 1/sqrt(3), rho0 = 1),
	// viscosity from the relaxation time, parameters from dimensionless numbers
	float p_from_rho(const float rho) const { return (rho-1.0f)/3.0f; }
	float rho_from_p(const float p) const { return 1.0f+3.0f*p; }
	float rho_laplace(const float sigma, const float R) const { return 6.0f*sigma/R; }                        // density offset of the Laplace pressure 2 sigma / R inside a sphere
	float rho_hydrostatic(const float f, const float z, const float h) const { return 3.0f*f*(h-z)+1.0f; }   // density at height z under a column of height h, force per volume f
	float nu_from_mu(const float mu, const float rho) const { return mu/rho; }
	float nu_from_tau(const float tau) const { return (tau-0.5f)/3.0f; }
	float nu_from_Re(const float Re, const float x, const float u) const { return x*u/Re; }
	float f_from_F(const float F, const float V) const { return F/V; }
	float f_from_g(const float g, const float rho) const { return rho*g; }
	float g_from_f(const float f, const float rho) const { return f/rho; }
	float u_from_Re(const float Re, const float x, const float nu) const { return Relations::velocity_for_reynolds(Re, x, nu); }
	float u_from_Re(const float Re, const float x, const float mu, const float rho) const { return Re*mu/(x*rho); }
	float u_from_Ma(const float Ma) const { return 0.57735027f*Ma; }
	float u_from_We(const float We, const float x, const float sigma, const float rho) const { return Relations::velocity_for_weber(We, x, sigma, rho); }
	float u_from_Fr(const float Fr, const float x, const float g) const { return Relations::velocity_for_froude(Fr, x, g); }
	float u_from_Ca(const float Ca, const float sigma, const float nu, const float rho) const { return Ca*sigma/(rho*nu); }
	float u_from_Ca(const float Ca, const float sigma, const float mu) const { return Ca*sigma/mu; }
	float u_from_f_Poiseuille_2D(const float f, const float rho, const float nu, const float R) const { return f*sq(R)/(2.0f*rho*nu); }
	float u_from_f_Poiseuille_3D(const float f, const float rho, const float nu, const float R) const { return f*sq(R)/(4.0f*rho*nu); }
	float u_from_f_Poiseuille_2D(const float Q, const float R) const { return Relations::poiseuille_centre_velocity_2D(Q, R); }
	float u_from_f_Poiseuille_3D(const float Q, const float R) const { return Relations::poiseuille_centre_velocity_3D(Q, R); }
	// Poiseuille flow: centre velocity u = f R^2 / (2 rho nu) between plates, f R^2 / (4 rho nu) in a pipe
	float f_from_u_Poiseuille_2D(const float u, const float rho, const float nu, const float R) const { return 2.0f*u*rho*nu/sq(R); }
	float f_from_u_Poiseuille_3D(const float u, const float rho, const float nu, const float R) const { return 4.0f*u*rho*nu/sq(R); }
	// rectangular duct of width w and height h: centre velocity u = (4 h^2 f / (pi^3 rho nu)) * sum over odd n of (1 - 1/cosh(n pi w / 2h)) sin(n pi/2) / n^3
	float f_from_u_rectangular_duct(const float w, const float h, const float rho, const float nu, const float u) const {
		const double kw = 0.5*3.14159265358979323846*(double)w/(double)h;
		double series = 0.0;
		for (int n = 45; n >= 1; n -= 2) series += (1.0-1.0/std::cosh(n*kw))/((double)n*n*n)*((n%4 == 1) ? 1.0 : -1.0); // smallest terms first
		const double pi3 = 3.14159265358979323846*3.14159265358979323846*3.14159265358979323846;
		return (float)(pi3*(double)rho*(double)nu*(double)u/(4.0*(double)h*(double)h)/series);
	}
	// Stokes flow past a sphere of radius R: drag 6 pi rho nu R u, also spread over a volume V
	float F_Stokes(const float rho, const float u, const float nu, const float R) const { return 6.0f*pif*rho*nu*R*u; }
	float f_Stokes(const float rho, const float u, const float nu, const float R, const float V) const { return F_Stokes(rho, u, nu, R)/V; }
	// creeping flow of far-field velocity u0 past a fixed sphere of radius R at the origin (FluidX3D-master/src/units.hpp:125-132):
	// u = u0 - (3R/4)(u0/r + x (u0.x)/r^3) - (R^3/4)(u0/r^3 - 3 x (u0.x)/r^5), p = -(3/2) rho nu R (u0.x)/r^3, rho = rho0 + 3p
	float3 u_Stokes(const float3& x, const float3& u0, const float R) const {
		const float r = length(x);
		if (r <= R) return float3(0.0f);
		const float ux = dot(u0, x), r3 = r*r*r, r5 = r3*r*r, R3 = R*R*R;
		return u0-(u0/r+x*(ux/r3))*(0.75f*R)-(u0/r3-x*(3.0f*ux/r5))*(0.25f*R3);
	}
	float rho_Stokes(const float3& x, const float3& u0, const float R, const float rho, const float nu) const {
		const float r = length(x);
		return r > R ? rho-4.5f*rho*nu*R*dot(u0, x)/(r*r*r) : rho;
	}

	// ---- the same relations for SI inputs (units.hpp:148-160; usable before the base units are set)
	float si_nu_from_si_mu(const float si_mu, const float si_rho) const { return si_mu/si_rho; }
	float si_nu_from_si_Re(const float si_Re, const float si_x, const float si_u) const { return si_x*si_u/si_Re; }
	float si_mu_from_si_nu(const float si_nu, const float si_rho) const { return si_nu*si_rho; }
	float si_f_from_si_g(const float si_g, const float si_rho) const { return si_rho*si_g; }
	float si_g_from_si_f(const float si_f, const float si_rho) const { return si_f/si_rho; }
	float si_u_from_si_Re(const float si_Re, const float si_x, const float si_nu) const { return Relations::velocity_for_reynolds(si_Re, si_x, si_nu); }
	float si_u_from_si_Re(const float si_Re, const float si_x, const float si_mu, const float si_rho) const { return si_Re*si_mu/(si_x*si_rho); }
	float si_u_from_si_We(const float si_We, const float si_x, const float si_sigma, const float si_rho) const { return Relations::velocity_for_weber(si_We, si_x, si_sigma, si_rho); }
	float si_u_from_si_Fr(const float si_Fr, const float si_x, const float si_g) const { return Relations::velocity_for_froude(si_Fr, si_x, si_g); }
	float si_u_from_si_h(const float si_h, const float si_g) const { return sqrt(2.0f*si_h*si_g); } // speed after a free fall from height h
	float si_u_Poiseuille_2D(const float si_Q, const float si_R) const { return Relations::poiseuille_centre_velocity_2D(si_Q, si_R); }
	float si_u_Poiseuille_3D(const float si_Q, const float si_R) const { return Relations::poiseuille_centre_velocity_3D(si_Q, si_R); }
};
extern Units units; // defined in lbm.cpp (FluidX3D-master/src/lbm.cpp:3)
