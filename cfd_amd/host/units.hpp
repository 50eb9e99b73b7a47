// units.hpp - SI <-> lattice unit conversion, the subset of the reference's Units class that the hot path's setups and
// the VTK export use (FluidX3D-master/src/units.hpp:5-161).
#pragma once
#include "utilities.hpp"

class Units {
	float unit_m = 1.0f, unit_kg = 1.0f, unit_s = 1.0f; // one lattice unit of length / mass / time in SI units
public:
	// fix the three base units from one length, one velocity and one density given in both unit systems
	void set_m_kg_s(const float x, const float u, const float rho, const float si_x, const float si_u, const float si_rho) {
		unit_m = si_x/x;
		unit_kg = si_rho/rho*cb(unit_m);
		unit_s = unit_m/(si_u/u);
	}
	void set_m_kg_s(const float m, const float kg, const float s) { unit_m = m; unit_kg = kg; unit_s = s; }
	// SI -> lattice
	float x(const float si_x) const { return si_x/unit_m; }
	ulong t(const float si_t) const { return to_ulong(si_t/unit_s); }
	float u(const float si_u) const { return si_u*unit_s/unit_m; }
	float rho(const float si_rho) const { return si_rho*cb(unit_m)/unit_kg; }
	float nu(const float si_nu) const { return si_nu*unit_s/sq(unit_m); }
	float f(const float si_f) const { return si_f*sq(unit_m*unit_s)/unit_kg; }
	float F(const float si_F) const { return si_F*sq(unit_s)/(unit_kg*unit_m); }
	// lattice -> SI
	float si_x(const uint x) const { return (float)x*unit_m; }
	float si_x(const float x) const { return x*unit_m; }
	float si_t(const ulong t) const { return (float)t*unit_s; }
	float si_u(const float u) const { return u*unit_m/unit_s; }
	float si_rho(const float rho) const { return rho*unit_kg/cb(unit_m); }
	float si_p(const float p) const { return p*unit_kg/(unit_m*sq(unit_s)); }
	float si_nu(const float nu) const { return nu*sq(unit_m)/unit_s; }
	float si_f(const float f) const { return f*unit_kg/sq(unit_m*unit_s); }
	float si_F(const float F) const { return F*unit_kg*unit_m/sq(unit_s); }
	float si_T(const float T) const { return T; }
	// dimensionless numbers and derived parameters
	float Re(const float x, const float u, const float nu) const { return x*u/nu; }
	float Ma(const float u) const { return u/0.57735027f; }
	float nu_from_tau(const float tau) const { return (tau-0.5f)/3.0f; }
	float nu_from_Re(const float Re, const float x, const float u) const { return x*u/Re; }
	float u_from_Re(const float Re, const float x, const float nu) const { return Re*nu/x; }
	float u_from_Ma(const float Ma) const { return 0.57735027f*Ma; }
	float f_from_u_Poiseuille_2D(const float u, const float rho, const float nu, const float R) const { return 2.0f*u*rho*nu/sq(R); }
	float f_from_u_Poiseuille_3D(const float u, const float rho, const float nu, const float R) const { return 4.0f*u*rho*nu/sq(R); }
	float F_Stokes(const float rho, const float u, const float nu, const float R) const { return 6.0f*pif*rho*nu*R*u; }
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
};
extern Units units; // defined in lbm.cpp (FluidX3D-master/src/lbm.cpp:3)
