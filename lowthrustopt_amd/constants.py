"""Physical constants of the reference, bit-identical (src/LowThrustOpt.jl:24-32)."""
MU = 0.012150585609624037   # mu_moon / (mu_moon + mu_planet)
DU = 384747.96285603708     # km per distance unit
TU = 375699.81732246041     # seconds per time unit
day = 86400.0               # seconds
r_moon = 1737.0
r_earth = 6378.0
mu_planet = 398600.4415
mu_moon = (MU * mu_planet) / (1 - MU)

# integrator ids (include/lto.h)
RK4, RKF78_FIXED, RKF78_ADAPTIVE, DOP853_ADAPTIVE = 0, 1, 2, 3
