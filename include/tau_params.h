/* tau_params.h — plain-C parameter blocks shared by the engine (libtaueng), the
 * thin C drivers and the CPU oracle.
 *
 * Every struct mirrors, field for field, the parameter block the reference
 * program keeps for the same simulator, so a reference `main` can hand its own
 * values across the C-ABI unchanged.  Citations are file:line in the reference
 * tree (seanwevans/fluid-sims).
 */
#ifndef TAU_PARAMS_H
#define TAU_PARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 3D two-temperature hypersonic Euler (tau_hypersonic_3d_cuda.cu:21-42) ---- */
typedef struct tau3d_params {
  int32_t nx, ny, nz;          /* GLOBAL grid (all ranks together)            */
  float dx, dy, dz;            /* 1/nx, 1/ny, 1/nz in the reference (:1535-1537) */
  float cfl;                   /* 0.3333 */
  float u_ref;                 /* asinh velocity scale, 10 */
  float R;                     /* gas constant, 10 */
  float gamma_floor;           /* 1.1 */
  float Twall;                 /* 0.02 */
  float tau_vib;               /* 2e-4 */
  float theta_v;               /* 0.2 */
  float sdf_cx, sdf_cy, sdf_cz;/* sphere centre, 0.5 */
  float sdf_r;                 /* 0.25 */
  float inflow_r, inflow_p;    /* 0.02, 0.02 */
  float inflow_u, inflow_v, inflow_w; /* 100, 0, 0 */
  int32_t sponge_n;            /* 24 */
  float sponge_strength;       /* 0.05 */
  int32_t sponge_out_n;        /* 24 */
  float sponge_out_strength;   /* 0.05 */
} tau3d_params;

/* Log-time clock of the 3D solver (tau_hypersonic_3d_cuda.cu:1635-1636, 1680-1704).
 * `t`,`d_tau` are the controller state BEFORE the next step; `dt`,`gain`,`maxs`
 * describe the step that was just taken. */
typedef struct tau3d_clock {
  float t;
  float d_tau;
  float dt;
  float gain;
  float maxs;
  int32_t step;
} tau3d_clock;

/* ---- Gray-Scott (tau_gray_scott.cu:43-61) ---- */
typedef struct taugs_params {
  int32_t nx, ny;
  float dx, dt;
  float Du, Dv, feed, kill;
} taugs_params;

/* ---- 5-point Laplacian viscosity passes ----
 * Burgers: tau_burgers.cu:490-525 (fields are asinh-encoded, u = u0*sinh(phi))
 * Shallow water: tau_shallow_water.cu:516-547 (plain u, v) */
typedef struct taulap_params {
  int32_t nx, ny;
  float dx, dy;
  float nu;
  float dt;
  float u0;      /* Burgers only: velocity scale of the asinh encoding */
} taulap_params;

/* ---- full Burgers / shallow-water programs (tau_burgers.cu:56-91, tau_shallow_water.cu:54-88).
 * One block for both; fields a program does not have are ignored.  Same names where they exist. */
typedef struct tauflow_params {
  int32_t nx, ny;          /* 512, 512 */
  float dx, dy;            /* 1, 1 */
  float nu;                /* Burgers 0.1 ; SW 0.001 */
  float u0;                /* Burgers: u = u0*sinh(phi), 1 */
  float g;                 /* SW 9.81 */
  float H0;                /* SW mean depth 1000 */
  float CFL;               /* Burgers 0.45 ; SW 0.5 */
  float tau0, t0, dtau;    /* 0, 1, 1 (log-time clock) */
  int32_t muscl;           /* Burgers --muscl */
  int32_t visc_substeps;   /* Burgers, 1 */
  int32_t oneD;            /* Burgers --colehopf (forces ny = 1) */
  /* initial field */
  float amp;               /* Burgers amp 1 ; SW bumpAmp 1 */
  float bsig;              /* Burgers bsig 16 ; SW bumpSigma 1 */
  float swirl;             /* Burgers 10 ; SW 1 */
  float rc;                /* Burgers rc 40 ; SW swirlRc 100 (cells) */
  float offx, offy;        /* Burgers 0,0 ; SW 100,100 */
  float asym;              /* Burgers 0 ; SW 10 */
  int32_t ck;              /* Cole-Hopf mode number 4 */
  float ca;                /* Cole-Hopf amplitude 0.5 */
} tauflow_params;

/* ---- 2D hypersonic Euler, GPU scheme (tau_hypersonic_cuda.cu:37-50, 1394-1409) ---- */
typedef struct tauh2_params {
  int32_t W, H;              /* compile-time 8192 x 1024 in the reference (:28-29) */
  double gamma;              /* 1.1 */
  double cfl;                /* 0.25 */
  double visc_nu;            /* 0.05 momentum */
  double visc_rho;           /* 0.05 */
  double visc_e;             /* 0.02 */
  double mach;               /* 25 */
  double geom_x0;            /* 125 */
  double geom_cy;            /* H/2 */
  double geom_rb;            /* H/12 */
  double geom_rn;            /* H/24 */
  double geom_theta;         /* pi/4 */
} tauh2_params;

/* ---- 2D WCSPH (tau_sph.cu:49-85; same names and defaults) ---- */
typedef struct tausph_params {
  int32_t N;            /* particle count, 1<<16 */
  float boxX, boxY;     /* 1, 1 */
  float dTau;           /* 1 */
  float t0;             /* 1 */
  float CFL;            /* 1 */
  float rho0;           /* 1 */
  float c0;             /* 1 */
  float gammaEOS;       /* 1 */
  float hMul;           /* 2 */
  float viscAlpha;      /* 0.25 */
  float gravity;        /* 9.81 */
  int32_t useVisc;      /* 1 */
  int32_t useGrav;      /* 1 */
  int32_t viscSub;      /* 1: sub-steps per step */
  int32_t seed;         /* 69420 */
  int32_t useXSPH;      /* 0: XSPH velocity smoothing after the integrate (--muscl / --xsph_eps, :81, 481-485) */
  float xsphEps;        /* 0.25 (:82) */
  int32_t rain;         /* rain inflow (:377-392, 706-716).  The reference's default is ON (:76) and it has no flag to
                           turn it off; the parameter default here is 0 because every recorded reference check-value
                           and BASELINE's SPH config are rain-off — the tau_sph driver sets 1 like the reference. */
} tausph_params;

/* ---- D2Q9 BGK lattice Boltzmann (tau_lbm.cu:43-55; same names and defaults) ---- */
typedef struct taulbm_params {
  int32_t nx, ny;          /* 512, 256 (the program clamps both to >= 16, :204-205) */
  int32_t obstacle;        /* 1: cylinder at (0.28 nx, 0.5 ny) */
  float tau;               /* 0.56: BGK relaxation time, viscosity = (tau - 1/2)/3 */
  float drive;             /* 1e-6: body-force-like x acceleration */
  float rho0;              /* 1 */
  float obstacle_radius;   /* 32 */
} taulbm_params;

#ifdef __cplusplus
}
#endif
#endif /* TAU_PARAMS_H */
