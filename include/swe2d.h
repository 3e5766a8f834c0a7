/*
 * swe2d.h - C ABI of the MI355X-native explicit 2D shallow-water stepper (libswe2d_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of thetisproject/thetis: the SSPRK33 time step of the
 * DG-P1 ('dg-dg', degree 1) 2D shallow water equations.  In the reference that path is
 *
 *   FlowSolver2d.iterate()                         thetis/solver2d.py:974-1144
 *     -> timestepper.advance(t, update_forcings)   thetis/rungekutta.py:949-952   (ERKGenericShuOsher)
 *        -> solve_stage(i): LinearVariationalSolver.solve() of  M k = dt R(U)     thetis/rungekutta.py:930-946
 *           R = ShallowWaterEquations.residual('all', U, U, ...)                  thetis/shallowwater_eq.py:922-928
 *
 * and the only contract FlowSolver2d has with a stepper is TimeIntegratorBase (thetis/timeintegrator.py:13-39):
 * ctor(equation, solution, fields, dt, options, bnd_conditions), initialize(solution), advance(t, update_forcings),
 * set_dt(dt); state is exchanged in place through `solution`.  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no exceptions cross the boundary: every function
 * returns 0 on success or a negative swe2d_status, and swe2d_last_error() gives the message.
 *
 * Host arrays use the reference's (Firedrake-side, [FD-assumed]) dof layout: cell c owns DG nodes 3c..3c+2 (= its
 * vertices in cell_vertices order); uv is (3N,2) row-major, eta is (3N).  Device layout is private (SoA planes).
 *
 * Threading: one host thread drives a handle; functions are not re-entrant per handle.  All work is enqueued on
 * the handle's HIP stream; functions that return data to host pointers synchronise that stream.
 */
#ifndef SWE2D_H
#define SWE2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWE2D_ABI_VERSION 12
#define SWE2D_MAX_MARKERS 16          /* boundary markers must be in 1..SWE2D_MAX_MARKERS-1 */

typedef enum {
    SWE2D_OK = 0,
    SWE2D_ERR_INVALID_ARGUMENT = -1,
    SWE2D_ERR_NO_DEVICE = -2,          /* no HIP device / HIP runtime error at init: the library never falls back to the CPU */
    SWE2D_ERR_HIP = -3,
    SWE2D_ERR_UNSUPPORTED = -4,
    SWE2D_ERR_NOT_FINITE = -5
} swe2d_status;

/* Boundary-condition kinds, bitmask.  Keys of bnd_functions['shallow_water'][marker]
 * (thetis/shallowwater_eq.py:232-296).  0 = closed (land) boundary. */
#define SWE2D_BC_ELEV 1
#define SWE2D_BC_UV   2
#define SWE2D_BC_UN   4
#define SWE2D_BC_FLUX 8
/* Function-valued boundary data (e.g. a tidal elevation field on an open boundary, updated by update_forcings): OR the
 * flag into `kind`; the value is then read from the nodal field given to swe2d_set_bc_field instead of values[]. */
#define SWE2D_BC_ELEV_FIELD 16
#define SWE2D_BC_UV_FIELD   32
#define SWE2D_BC_UN_FIELD   64
#define SWE2D_BC_FLUX_FIELD 128

/* Nodal coefficient fields, `fields` dict of get_swe_timestepper (thetis/solver2d.py:547-559). */
typedef enum {
    SWE2D_FIELD_CORIOLIS = 0,              /* options.coriolis_frequency      shallowwater_eq.py:623-634 */
    SWE2D_FIELD_ATMOSPHERIC_PRESSURE = 1,  /* options.atmospheric_pressure    shallowwater_eq.py:658-663 */
    SWE2D_FIELD_MOMENTUM_SOURCE = 2,       /* options.momentum_source_2d (3N,2) shallowwater_eq.py:805-811 */
    SWE2D_FIELD_VOLUME_SOURCE = 3,         /* options.volume_source_2d        shallowwater_eq.py:825-831 */
    SWE2D_FIELD_WIND_STRESS = 4,           /* options.wind_stress (3N,2)      shallowwater_eq.py:643-649 */
    /* spatially varying drag coefficients (nodal DG values); a field replaces the scalar of the same name, the same
     * exclusions apply (at most one of quadratic / Manning / Nikuradse, shallowwater_eq.py:686-696) */
    SWE2D_FIELD_LINEAR_DRAG = 5,           /* options.linear_drag_coefficient as a Function */
    SWE2D_FIELD_QUADRATIC_DRAG = 6,        /* options.quadratic_drag_coefficient as a Function */
    SWE2D_FIELD_MANNING_DRAG = 7,          /* options.manning_drag_coefficient as a Function */
    SWE2D_FIELD_NIKURADSE = 8,             /* options.nikuradse_bed_roughness as a Function */
    SWE2D_FIELD_COUNT = 9
} swe2d_field;

/* Scalar coefficients (Constants in the reference). */
typedef enum {
    SWE2D_SCALAR_LINEAR_DRAG = 0,          /* options.linear_drag_coefficient     shallowwater_eq.py:734-740 */
    SWE2D_SCALAR_QUADRATIC_DRAG = 1,       /* options.quadratic_drag_coefficient  shallowwater_eq.py:683,699-700 */
    SWE2D_SCALAR_MANNING_DRAG = 2,         /* options.manning_drag_coefficient    shallowwater_eq.py:685-688 */
    SWE2D_SCALAR_NORM_SMOOTHER = 3,        /* options.norm_smoother               shallowwater_eq.py:700 */
    SWE2D_SCALAR_NIKURADSE = 4,            /* options.nikuradse_bed_roughness     shallowwater_eq.py:692-700 (kappa = 0.4) */
    SWE2D_SCALAR_COUNT = 5
} swe2d_scalar;

/* Library options: how the path is carried out, never what it computes (every setting gives the same bits; the parity tests
 * force each in turn).  Read by the library from the handle only - it never looks at the process environment; the THETIS_AMD_*
 * variables of the same names are a convenience of the Python binding (thetis_amd/_lib.py: OPTION_ENV), applied once after
 * swe2d_create.  value -1 = the library's own rule.  No reference counterpart (Firedrake's analogue is the
 * solver_parameters / PyOP2 configuration a script may pass, thetis/options.py:145-152). */
typedef enum {
    SWE2D_OPT_FUSED_STAGES = 0,   /* stages of a step in one launch by overlapped tiles (csrc/swe2d_fuse.h): -1 from 250 k triangles
                                     when the numbering gives compact tiles - and all three stages in one launch on whole meshes
                                     without source terms (swe2d_fused_triple_info): from 131 073 cells where the caller handed in
                                     patches (swe2d_fused_set_triple_tiles), from 2.5 M otherwise; 0 never; 1 the pair on every mesh the
                                     kernel covers; 2 the pair by the size rule, never the triple; 3 the triple on every whole mesh */
    SWE2D_OPT_FLOW = 1,           /* swe2d_advance takes the dataflow stage loop (csrc/swe2d_flow.h) where it applies: -1 / 1 yes, 0 no */
    SWE2D_OPT_FLOW_WD = 2,        /* ... also with wetting-drying: -1 / 1 yes, 0 no */
    SWE2D_OPT_BND_INLINE = 3,     /* triangle stage kernels: boundary facets from registers (1) or by the epilogue that reloads them (0);
                                     -1: inline, with wetting-drying the epilogue.  0 also keeps the fused / dataflow kernels away */
    SWE2D_OPT_LDSX = 4,           /* in-wave neighbour traces through LDS: -1 in launches beyond the Infinity Cache, 0 / 1 forced */
    SWE2D_OPT_ALTERNATE = 5,      /* alternating launch direction: -1 in launches beyond the Infinity Cache, 0 / 1 forced */
    SWE2D_OPT_COMPACT_IDX = 6,    /* 16-B connectivity records: -1 / 1 in launches of >= 250 k cells, 0 never, 2 in every launch */
    SWE2D_OPT_VISC_FUSION = 7,    /* triangles: viscosity inside the stage kernel (-1 / 1) or as a separate pass (0) */
    SWE2D_OPT_WALL_FAST = 8,      /* closed walls on a path of their own in the boundary code (-1 / 1) or through the general one (0) */
    SWE2D_OPT_FLOW_POLL = 9,      /* granule loads per polling trip of the dataflow kernel: -1 by the blocks' rim facets, else 3 ... 9 */
    SWE2D_OPT_FLOW_CAPACITY = 10, /* resident 64-cell blocks the dataflow kernel may assume: -1 what the device holds (tests force less) */
    SWE2D_OPT_FLOW_TIMEOUT_MS = 11, /* bound of every wait inside the dataflow kernel, default 2000 */
    SWE2D_OPT_P2P_TIMEOUT_MS = 12,  /* bound of the peer-to-peer waits, default 5000 */
    SWE2D_OPT_P2P_ZONE = 13,      /* landing zone memory: -1 first that works, 1 uncached, 2 fine-grained, 3 ordinary device memory */
    SWE2D_OPT_ROCTX = 14,         /* 1: ROCTx ranges around the entry points that advance the state (rocprofv3 --marker-trace) */
    SWE2D_OPT_COUNT = 15
} swe2d_option;

/* Mesh = what FlowSolver2d(mesh2d, bathymetry_2d) receives (thetis/solver2d.py:81-147) flattened to arrays.
 * Triangles (nodes_per_cell == 3, DG-P1) or convex quadrilaterals (nodes_per_cell == 4, DQ-1), counter-clockwise.  A mesh of
 * parallelograms (every quadrilateral mesh the reference builds itself) takes the kernels with a constant Jacobian and the tensor
 * mass inverse; one cell that is not a parallelogram selects the general bilinear kernels for the whole mesh (Jacobian per
 * quadrature point, 4 x 4 mass solve per cell, mass-weighted cell means; thetis/solver2d.py:340-345 accepts any quadrilateral
 * mesh) - every option of the path runs on either kind.
 * Local facet f joins local vertices f and (f+1)%nodes_per_cell; all [..][3] shapes below read [..][nodes_per_cell].
 * In a multi-GPU partition the first n_owned cells are updated by this handle and cells n_owned..n_cells-1 are
 * ghost cells (one layer, facet-adjacent) whose state arrives through swe2d_halo_*. */
typedef struct {
    int32_t n_cells;
    int32_t n_owned;                   /* == n_cells on a single device */
    int32_t n_vertices;
    int32_t nodes_per_cell;            /* 3 or 4 */
    const int32_t *cell_vertices;      /* [n_cells][3] */
    const double  *vertex_xy;          /* [n_vertices][2] */
    const int32_t *cell_neighbours;    /* [n_cells][3]  >=0: neighbour cell, <0: -(boundary marker) */
    const int8_t  *cell_neighbour_facets; /* [n_cells][3] local facet id of the shared facet inside the neighbour */
    const double  *bathymetry;         /* [n_vertices] CG-P1 bathymetry, positive down (fields.bathymetry_2d) */
    const double  *boundary_len;       /* [SWE2D_MAX_MARKERS] total length per marker (utility.py:821-832), may be NULL on
                                          a single device (computed); REQUIRED for partitions (global lengths) */
} swe2d_mesh;

/* The ModelOptions2d entries the path reads (thetis/options.py:583-733). */
typedef struct {
    double  g_grav;                              /* physical_constants['g_grav'] = 9.81 */
    double  dt;                                  /* options.timestep */
    int32_t use_nonlinear_equations;             /* default 1 */
    int32_t use_lax_friedrichs_velocity;         /* default 1 */
    double  lax_friedrichs_velocity_scaling_factor; /* default 1.0 */
    int32_t device_id;                           /* HIP device ordinal */
} swe2d_params;

typedef struct swe2d_handle swe2d_handle;

/* version / capability */
int  swe2d_abi_version(void);
/* How the triangle kernels read the connectivity: out[0] = 1 when from the 16-B records (neighbour and vertex ids as differences
 * to the cell's own, csrc/swe2d_kernels.h swe_conn_pack; in launches of >= 250 k cells, where it pays;
 * SWE2D_OPT_COMPACT_IDX = 0 keeps the 24-B records everywhere, = 2 takes the 16-B records in every launch), out[1] = the number
 * of cells whose differences did not fit and which read the 24-B record after all.  Results do not depend on it. */
int  swe2d_connectivity_info(swe2d_handle *h, int32_t out[2]);
/* Stages 1 and 2 of a step in ONE launch by overlapped tiles (csrc/swe2d_fuse.h: 192 interior cells + their ring per workgroup,
 * the first stage's result never leaves the chip), stage 3 as a stage launch: what swe2d_advance and swe2d_advance_coupled do from
 * 250 k triangles without wetting-drying or viscosity (source terms are covered), when the cell numbering gives compact tiles, and
 * what swe2d_solve_stage_pair_cells does on a partition of that size (SWE2D_OPT_FUSED_STAGES = 0:
 * never, = 1: on every such mesh).  Same results bit for bit; the intermediate stage_sol[0] = U(1) then never reaches the state buffers
 * (swe2d_get_stage_state(h, 0) after such a step returns SWE2D_ERR_UNSUPPORTED, not a stale buffer).
 * out[0] = 1 when swe2d_advance would take it now (builds the tile tables on first use), out[1] = tiles, out[2] = ring cells
 * (cells evaluated redundantly in stage 1), out[3] = cells. */
int  swe2d_fused_pair_info(swe2d_handle *h, int32_t out[4]);
/* The tiles are cut from consecutive cells of an ORDER (default: the numbering of swe2d_mesh, NULL restores it): a partition whose
 * ghost layers are appended to its numbering layer by layer passes one in which every ghost cell sits next to the owned cells it
 * touches (cf. swe2d_flow_set_order).  Results do not depend on it.  Drops tile tables built before; synchronises the stream. */
int  swe2d_fused_set_order(swe2d_handle *h, const int32_t *cells_in_tile_order);
/* ALL three stages of a step in one launch (by itself from 2.5 M triangles, from 131 073 with the caller's patches - see SWE2D_OPT_FUSED_STAGES; = 3 forces it; csrc/swe2d_fuse.h swe_fuse123_kernel: tiles of interior +
 * two rings, U(1) and U(2) stay on chip, U(3) goes to the second state buffer and the two change places - so not inside a stream
 * capture, where swe2d_advance keeps the fused pair): whole meshes; out[0] = 1 when swe2d_advance would take it now (builds
 * the tile tables), out[1] = tiles, out[2] / out[3] = cells of the first / second rings (stage 1 is evaluated on interior + both,
 * stage 2 on interior + the first).  Same results bit for bit. */
int  swe2d_fused_triple_info(swe2d_handle *h, int32_t out[4]);
/* The two-ring tiles alone may be cut from an order of their own, with positions at which a tile must begin (tile_starts, n_starts
 * of them; NULL / 0: none): compact patches sized for interior + two rings (thetis_amd/ordering.py triple_tile_order: 12 x 7 quads of
 * a RectangleMesh) instead of as many consecutive cells as fit.  cells_in_tile_order = NULL: the order of swe2d_fused_set_order.
 * Results do not depend on it.  Replaces nothing in the reference (Firedrake's PyOP2 has no tiling to steer). */
int  swe2d_fused_set_triple_tiles(swe2d_handle *h, const int32_t *cells_in_tile_order, const int32_t *tile_starts, int32_t n_starts);
/* A PARTITION's whole step in one launch (round 6, last): stage 3 on cells [0, cell_end), the last of the step's three shrinking ranges
 * (thetis_amd/partition.py stage_range); stages 1 and 2 are evaluated on the tiles' supersets of theirs and never leave the chip.  The
 * result goes to the second state buffer and the two change places: inside a stream capture call it an EVEN number of times per
 * captured sequence (the pointers a replay uses are those of the capture; an odd number is reported as SWE2D_ERR_UNSUPPORTED by the
 * next swe2d_synchronize / swe2d_get_stage_state / swe2d_solve_step_cells outside the capture), and build the tables before
 * (swe2d_fused_step_info).
 * Cells of the state beyond cell_end hold stale values afterwards (ghost cells the next exchange rewrites).  SWE2D_ERR_UNSUPPORTED
 * where the kernel does not cover the handle (quadrilaterals, wetting-drying, viscosity, source terms unless forced).
 * swe2d_fused_step_info: out[0] = 1 when the caller should take it (patches handed in with swe2d_fused_set_triple_tiles and more
 * cells than the dataflow kernel holds, or SWE2D_OPT_FUSED_STAGES = 3), out[1..3] as swe2d_fused_triple_info.  Same bits as
 * swe2d_solve_stage_cells x 3.  Replaces: one ERKGenericShuOsher.advance on a rank's cells (thetis/rungekutta.py:949-952). */
int  swe2d_solve_step_cells(swe2d_handle *h, int32_t cell_end);
int  swe2d_fused_step_info(swe2d_handle *h, int32_t out[4]);
int  swe2d_device_count(void);                                   /* number of visible HIP devices, <0 on error */

/* Shu-Osher coefficients the stage kernels use (host-only, needs no device): stage i computes
 * U_{i+1} = beta[i]*k_i + alpha0[i]*U_0 + alpha_in[i]*U_i  (alpha_in[0] multiplies U_0 itself).  Values are the output of
 * butcher_to_shuosher_form (thetis/rungekutta.py:13-87) for SSPRK33Abstract (rungekutta.py:342-346). */
void swe2d_ssprk33_coefficients(double alpha0[3], double alpha_in[3], double beta[3]);

/* lifetime: replaces ERKGenericShuOsher.__init__/update_solver (rungekutta.py:877-924).
 * One handle = one device.  The kernels address a group of nodes_per_cell SoA planes through one 4 GiB buffer resource with
 * 32-bit offsets: nodes_per_cell * n_cells * 8 bytes must stay below 2^32 (about 178 M triangles / 134 M quadrilaterals per
 * device), larger meshes return SWE2D_ERR_UNSUPPORTED and have to be partitioned (one handle per part). */
int  swe2d_create(const swe2d_mesh *mesh, const swe2d_params *params, swe2d_handle **out);
/* Partitions of ONE quadrilateral mesh must all take the same kernel family: the parallelogram kernels and the general bilinear
 * ones differ in the last bits on the same parallelogram cell, and a ghost cell must repeat its owner's arithmetic.  A handle
 * whose own cells are all parallelograms while the global mesh has a general cell is told so here (on = 1: general kernels;
 * on = 0: back to what the handle's own cells allow).  Call before the first step. */
int  swe2d_set_general_quadrilaterals(swe2d_handle *h, int on);
void swe2d_destroy(swe2d_handle *h);
const char *swe2d_last_error(const swe2d_handle *h);             /* h may be NULL: error of the last failed create */
/* library options (swe2d_option above); takes effect from the next call on */
int  swe2d_set_option(swe2d_handle *h, int option, int value);
int  swe2d_get_option(swe2d_handle *h, int option, int *value);

/* state: the mixed Function solution_2d = (uv_2d, elev_2d) (solver2d.py:410-413); host pointers */
int  swe2d_set_state(swe2d_handle *h, const double *uv, const double *eta);
int  swe2d_get_state(swe2d_handle *h, double *uv, double *eta);
/* the stage solution the reference assigns to `solution` after solve_stage(i_stage) (rungekutta.py:930-946): i_stage 0, 1 read
 * the buffers holding U1, U2; i_stage 2 (= swe2d_get_state) the step result.  The reference's stage_sol[i] always is what stage i
 * left; here the fused stage pair keeps U1, the dataflow kernel U1 and U2, on chip: when the step made last did not leave the asked
 * stage solution in memory (or no stage has run since swe2d_set_state / a restore) the call returns SWE2D_ERR_UNSUPPORTED - drive
 * the step with swe2d_solve_stage to read intermediate stages */
int  swe2d_get_stage_state(swe2d_handle *h, int i_stage, double *uv, double *eta);
/* Save (restore = 0) / bring back (restore = 1) the time-stepping state - the step result and every tracer - in a device-side
 * copy, exactly, without the host: for steps that have to be undone (graph capture warm-ups, verification replays, benchmarks).
 * No reference counterpart (there `solution.assign(saved)` does it; here swe2d_get_state + swe2d_set_state is exact only without
 * wetting-drying: with it the device carries the displaced depth D = (H + sqrt(H^2 + alpha^2))/2 instead of eta, which it hands
 * out and takes in through the closed forms of thetis/utility.py:975-996 - the identity up to rounding, not bit for bit).
 * Enqueued on the handle's stream. */
int  swe2d_state_snapshot(swe2d_handle *h, int restore);         /* slot 0 */
/* ... in one of SWE2D_SNAPSHOT_SLOTS independent slots: a caller that keeps a long-lived copy (the start of a verification window)
 * and needs short-lived ones in between (graph capture warm-ups) gives each its own */
#define SWE2D_SNAPSHOT_SLOTS 2
int  swe2d_state_snapshot_slot(swe2d_handle *h, int slot, int restore);

/* TimeIntegrator.set_dt (timeintegrator.py:70-73) */
int  swe2d_set_dt(swe2d_handle *h, double dt);

/* bnd_functions['shallow_water'][marker] = {...} with constant values (shallowwater_eq.py:232-272).
 * kind = bitmask of SWE2D_BC_*; values = {elev, u, v, un, flux}.  May be called between stages
 * (update_forcings, rungekutta.py:933-934). */
int  swe2d_set_bc(swe2d_handle *h, int marker, int kind, const double values[5]);

/* Function-valued boundary data of ONE marker (bnd_functions[...][marker][key] = Function): nodal DG values of the whole
 * mesh in the host layout, which = 0: elevation (kN), 1: velocity (kN,2), 2: normal velocity (kN), 3: flux (kN).  Only the
 * nodes of boundary facets carrying `marker` are copied (stored per facet, so the two boundaries meeting at a corner cell
 * keep their own values).  Select the field with the SWE2D_BC_*_FIELD bit in swe2d_set_bc.  May be called between stages. */
int  swe2d_set_bc_field(swe2d_handle *h, int which, int marker, const double *nodal);
/* The same data as a COMPACT list - what update_forcings should cost per call (a tidal elevation Function re-evaluated at
 * t + c_i dt, rungekutta.py:933-934: a few KB instead of the whole nodal field): entry t is boundary facet facets[t] of cell
 * cells[t] (caller's cell numbering as passed to swe2d_create), values[t][j][c] the value at the facet's first (j = 0) and
 * second (j = 1) node, component c (2 components for which = 1, else 1).  Facets not listed keep their values. */
int  swe2d_set_bc_facets(swe2d_handle *h, int which, int32_t n_facets, const int32_t *cells, const int32_t *facets,
                         const double *values);

/* bnd_functions['shallow_water'][marker]['drag'] = C_D (BoundaryDragTerm, shallowwater_eq.py:704-725); negative: none */
int  swe2d_set_boundary_drag(swe2d_handle *h, int marker, double drag_coefficient);

/* coefficient fields: nodal DG-P1 values (3N) [(3N,2) for the momentum source and the wind stress] or NULL to switch the term off */
int  swe2d_set_field(swe2d_handle *h, int field, const double *nodal);
/* a CONTINUOUS P1 coefficient given per vertex (numbering of swe2d_mesh.vertex_xy), [n_vertices] or [n_vertices][2] for the
 * vector fields: the injection into the DG nodes happens on the device.  A time-dependent wind / pressure Function then costs
 * one value per vertex per update instead of one per DG node. */
int  swe2d_set_field_vertex(swe2d_handle *h, int field, const double *vertex_values);
/* scalar coefficients; a negative value switches the term off (norm_smoother: >= 0) */
int  swe2d_set_scalar(swe2d_handle *h, int which, double value);

/* options.use_wetting_and_drying + options.wetting_and_drying_alpha (thetis/options.py:872-884), alpha given at the mesh
 * vertices (constant or the P1 field of set_wetting_and_drying_alpha, solver2d.py:251-303).  The reference cannot run
 * SSPRK33 with wetting-drying; this enables the build's own explicit nodal formulation of the same
 * displaced depth: the continuity equation advances zeta = D - h, every stage ends with a positivity
 * limiter on the nodal depths (D >= 0.1 alpha) and a relaxation of the velocity on dry ground.  Enable it BEFORE
 * swe2d_set_state: the state is then brought to the admissible set (nodal depths through the limiter).  swe2d_diagnostics
 * reports int D dx, min D in slots 2, 3; swe2d_tendency returns the raw tendencies of (u, v, zeta). */
int  swe2d_set_wetting_and_drying(swe2d_handle *h, int enable, const double *alpha_vertex);

/* ERKGenericShuOsher.advance (rungekutta.py:949-952) repeated n_steps times, forcings constant in time.
 * Asynchronous: returns after enqueueing. */
int  swe2d_advance(swe2d_handle *h, int n_steps);
/* ERKGenericShuOsher.solve_stage(i_stage) (rungekutta.py:930-946); i_stage = 0,1,2 in order. */
int  swe2d_solve_stage(swe2d_handle *h, int i_stage);
/* timeintegrator.ForwardEuler.advance (thetis/timeintegrator.py:115-165; 'ForwardEuler' in the steppers table,
 * solver2d.py:664) x n_steps:  U <- U + dt M^-1 R(U) - the first Shu-Osher stage followed by a buffer swap */
int  swe2d_advance_forward_euler(swe2d_handle *h, int n_steps);
/* n_steps steps bracketed by HIP events on the handle's stream; *ms_total = elapsed GPU time,
 * *ms_kernel_avg = mean duration of one stage kernel launch (events around every launch when per_launch != 0). */
int  swe2d_advance_timed(swe2d_handle *h, int n_steps, int per_launch, float *ms_total, float *ms_kernel_avg);
int  swe2d_synchronize(swe2d_handle *h);

/* parity hook: tendency k = M^-1 (dt R(U)) of the current state (what solver.solve() leaves in `tendency`,
 * rungekutta.py:940), host layout as the state. */
int  swe2d_tendency(swe2d_handle *h, double *k_uv, double *k_eta);

/* print_state norms + VolumeConservation2DCallback (solver2d.py:955-956, callback.py:350-364, utility.py:421-425):
 * out = { int eta^2 dx, int |u|^2 dx, int (eta+h) dx, min nodal (h+eta) } over the owned cells
 * (sums, not roots, so that partitions can be added). */
int  swe2d_diagnostics(swe2d_handle *h, double out[4]);
/* The same integrals as order-independent sums, for runs partitioned over several handles / ranks (the reference all-reduces its
 * diagnostics over the MPI ranks, thetis/callback.py:478-482; a floating-point all-reduce would make the printed norms and the
 * conservation checks depend on the partition in their last digits): every cell's contribution is split exactly into four
 * signed 38-bit limbs of units 2^40, 2^2, 2^-36, 2^-74, 2^-112, 2^-150 (what a term has below 2^-150 ~ 7e-46 is dropped) and the limbs are
 * summed as integers.  limbs[6 q + j] = limb j of
 * quantity q (int eta^2, int |u|^2, int (eta+h)); add the limbs of all partitions (int64, any order), then
 * swe2d_sum_limbs_to_double rounds a total to the nearest double.  swe2d_diagnostics returns exactly that for its own handle, so
 * one handle over the whole mesh and N handles over its partitions give identical doubles. */
int  swe2d_diagnostics_limbs(swe2d_handle *h, int64_t limbs[18], double *min_depth);
double swe2d_sum_limbs_to_double(const int64_t limbs[6]);

/* ---- SIPG horizontal viscosity: HorizontalViscosityTerm (thetis/shallowwater_eq.py:554-616), fields['viscosity_h'] =
 * options.horizontal_viscosity (solver2d.py:551).  nu is a constant (nu_vertex == NULL) or a continuous P1 field given per
 * vertex; sipg_factor = options.sipg_factor (options.py:730); the two flags are options.use_grad_div_viscosity_term and
 * use_grad_depth_viscosity_term (options.py:597-606).  Dirichlet boundary terms follow the velocity-type boundary
 * conditions set with swe2d_set_bc (:584-609).  With wetting and drying the depth of the grad-depth term is the displaced depth
 * and the dry-ground relaxation acts on the whole new velocity (viscous share included).  enable = 0 switches the
 * term off (viscosity_h None, :559-560). */
int  swe2d_set_viscosity(swe2d_handle *h, int enable, const double *nu_vertex, double nu_const, double sipg_factor,
                         int use_grad_div_viscosity_term, int use_grad_depth_viscosity_term);

/* ---- 2D tracers (thetis/tracer_eq_2d.py, non-conservative form) and the vertex-based P1DG limiter -----------------
 * A tracer is a scalar DG-P1 field (3N nodal values, same node layout as eta) advected by the CURRENT shallow-water
 * velocity of the handle.  Single-device handles only for now. */
int  swe2d_tracer_add(swe2d_handle *h, int *tracer_id);          /* options.add_tracer_2d (options.py:945-983) */
/* options.use_lax_friedrichs_tracer, lax_friedrichs_tracer_scaling_factor, tracer_advective_velocity_factor */
int  swe2d_tracer_set_options(swe2d_handle *h, int use_lax_friedrichs_tracer, double lax_friedrichs_tracer_scaling_factor,
                              double tracer_advective_velocity_factor);
int  swe2d_tracer_set_state(swe2d_handle *h, int tracer_id, const double *nodal);
int  swe2d_tracer_get_state(swe2d_handle *h, int tracer_id, double *nodal);
/* bnd_functions['tracer_2d'][marker] = {'value': c}; has_value = 0 restores the default boundary term
 * c (u.n) phi (tracer_eq_2d.py:177-191).  Velocity-type keys: swe2d_tracer_set_bc_velocity. */
int  swe2d_tracer_set_bc(swe2d_handle *h, int tracer_id, int marker, int has_value, double value);
/* external velocity of the tracer's boundary dict on `marker` (tracer_eq_2d.py:70-110): kind 0 = none (uv_ext = uv_in),
 * 1 = 'uv': (u, v), multiplied by tracer_advective_velocity_factor, 2 = 'un': normal velocity u (v unused),
 * 3 = 'flux': volume flux u out of the domain, uv_ext = factor * u / (H(elev_in) * boundary_len) n,
 * 4 = 'flux' with a constant 'elev' in the dict: the same with H(v) */
int  swe2d_tracer_set_bc_velocity(swe2d_handle *h, int tracer_id, int marker, int kind, double u, double v);
/* the same with Function-valued entries in compact form (tracer_eq_2d.py:100-109): values[n_facets][2] (kind 1 'uv':
 * [n_facets][2][2]) = the entry at the first and second node of boundary facet `facets[i]` of cell `cells[i]`; elev: the
 * constant 'elev' of a 'flux' entry (kind 4), ignored otherwise */
int  swe2d_tracer_set_bc_velocity_facets(swe2d_handle *h, int tracer_id, int marker, int kind, double elev, int32_t n_facets,
                                         const int32_t *cells, const int32_t *facets, const double *values);
/* Function-valued 'value' on `marker`: nodal DG values of the whole mesh in the host layout (kN); only cells with a
 * boundary facet carrying `marker` are copied (all their nodes: the diffusive boundary term uses the cell gradient) */
int  swe2d_tracer_set_bc_field(swe2d_handle *h, int tracer_id, int marker, const double *nodal);
/* compact form: values[t][i] = external value at node i of cell cells[t], stored for its boundary facet facets[t] */
int  swe2d_tracer_set_bc_facets(swe2d_handle *h, int tracer_id, int32_t n_facets, const int32_t *cells, const int32_t *facets,
                                const double *values);     /* then select them: swe2d_tracer_set_bc(..., has_value = 2, 0) */
int  swe2d_tracer_set_source(swe2d_handle *h, int tracer_id, const double *nodal);   /* SourceTerm tracer_eq_2d.py:281-298 */
/* options.tracer[label].use_conservative_form (options.py:543): the tracer field is the depth-integrated q = H*T and the
 * stage kernels evaluate ConservativeHorizontalAdvectionTerm / ConservativeSourceTerm (tracer_eq_2d.py:325-437) */
int  swe2d_tracer_set_conservative(swe2d_handle *h, int tracer_id, int use_conservative_form);
/* SIPG horizontal diffusion: HorizontalDiffusionTerm (tracer_eq_2d.py:226-278), fields['diffusivity_h-<label>'] =
 * options.tracer[label].diffusivity (solver2d.py:588); constant or per-vertex P1; sipg_factor_tracer options.py:732 */
int  swe2d_tracer_set_diffusivity(swe2d_handle *h, int tracer_id, int enable, const double *mu_vertex, double mu_const,
                                  double sipg_factor_tracer);
/* boundary term of the diffusion operator on `marker` (tracer_eq_2d.py:264-277): kind 0 = none (no boundary dict),
 * 1 = prescribed 'diff_flux' (-phi*diff_flux); otherwise -phi mu grad(c_up).n with c_up = s c + (1-s) c_ext, s the upwind
 * switch: 2 = constant 'value' (grad c_ext = 0), 3 = boundary dict without 'value' (c_ext = c), 4 = Function 'value' */
#define SWE2D_DIFF_BC_NONE 0
#define SWE2D_DIFF_BC_DIFF_FLUX 1
#define SWE2D_DIFF_BC_UPWIND 2
#define SWE2D_DIFF_BC_GRAD_IN 3
#define SWE2D_DIFF_BC_VALUE_FIELD 4
int  swe2d_tracer_set_diffusion_bc(swe2d_handle *h, int tracer_id, int marker, int kind, double diff_flux);
int  swe2d_tracer_solve_stage(swe2d_handle *h, int tracer_id, int i_stage);          /* rungekutta.py:930-946 for the tracer */
int  swe2d_tracer_tendency(swe2d_handle *h, int tracer_id, double *k_nodal);
int  swe2d_tracer_forward_euler(swe2d_handle *h, int tracer_id);                     /* one ForwardEuler step of the tracer */
/* partitions (one handle per GPU): the tracer stage on a local cell range, the limiter on cells [0, cell_end) with means
 * and vertex bounds taken over every local cell (the ghost layers must hold the neighbours' unlimited values), and the
 * tracer's part of the halo exchange (same cell lists as swe2d_halo_setup, nodes_per_cell doubles per cell) */
int  swe2d_tracer_solve_stage_cells(swe2d_handle *h, int tracer_id, int i_stage, int32_t cell_begin, int32_t cell_end);
/* ForwardEuler for a tracer on a partition: swe2d_tracer_solve_stage_cells(id, 0, begin, end) is the step from tracer buffer 0
 * into buffer 1 on a cell range; after the last range swe2d_tracer_swap_buffers makes buffer 1 the tracer (cf. swe2d_forward_euler_cells) */
int  swe2d_tracer_swap_buffers(swe2d_handle *h, int tracer_id);
int  swe2d_tracer_limit_cells(swe2d_handle *h, int tracer_id, int32_t cell_end);
int  swe2d_tracer_halo_pack(swe2d_handle *h, int tracer_id, int i_buffer, double *send_buf_dev);
int  swe2d_tracer_halo_unpack(swe2d_handle *h, int tracer_id, int i_buffer, const double *recv_buf_dev);         /* parity hook */
/* VertexBasedP1DGLimiter (thetis/limiter.py:48-198): topology of the mesh vertices (periodic meshes identify them);
 * optional - defaults to cell_vertices. */
int  swe2d_limiter_setup(swe2d_handle *h, int32_t n_topo_vertices, const int32_t *cell_topo_vertices);
int  swe2d_tracer_limit(swe2d_handle *h, int tracer_id);                             /* limiter.apply(field) */
/* out = { int T*H dx (comp_tracer_mass_2d, utility.py:437-445), int T dx, min nodal T, max nodal T } */
int  swe2d_tracer_diagnostics(swe2d_handle *h, int tracer_id, double out[4]);
/* limb sums (see swe2d_diagnostics_limbs) of { int T*H dx, int T dx } + { min, max } of the owned cells */
int  swe2d_tracer_diagnostics_limbs(swe2d_handle *h, int tracer_id, int64_t limbs[12], double minmax[2]);
/* GeneralCoupledTimeIntegrator2D.advance (coupled_timeintegrator_2d.py:93-113) x n_steps: SWE step (unless tracer_only),
 * then every tracer with the updated velocity, then the limiter (once per step) */
int  swe2d_advance_coupled(swe2d_handle *h, int n_steps, int tracer_only, int use_limiter);

/* profiling aid: n_times streaming copies of the 9 state planes (9*stride doubles read + written, 8 B per lane) to calibrate
 * the FETCH_SIZE / WRITE_SIZE counters on a known byte count; does not change the state */
int  swe2d_debug_calibration_copy(swe2d_handle *h, int n_times);

/* ---- multi-GPU plumbing (one process per GPU; the exchange itself is done by the host with RCCL) ----
 * Replaces PyOP2's per-par_loop halo exchange [FD-assumed] by ONE exchange per time step: a partition carries three layers
 * of ghost cells (cells n_owned..n_cells-1, layer by layer); stage i is run on cells [0, n_owned + layers still needed),
 * see thetis_amd/partition.py.  send_cells: local ids of owned cells whose state peers need, grouped by peer;
 * recv_cells: local ids of the ghost cells in the order the peers' messages deliver them.  Buffers are device pointers
 * owned by the caller (cell-major: 3k doubles u0..u(k-1) v0.. e0.. per cell, [n][3k], k = nodes_per_cell, so per-peer
 * segments are contiguous).  i_buffer selects the state buffer (0 = the step result / stage-1 input).  On a handle without ghost
 * cells (n_owned = n_cells) recv_cells may name any cell: pack + unpack then copy between cells of the same mesh. */
int  swe2d_halo_setup(swe2d_handle *h, int32_t n_send, const int32_t *send_cells, int32_t n_recv, const int32_t *recv_cells);
int  swe2d_halo_pack(swe2d_handle *h, int i_buffer, double *send_buf_dev);
int  swe2d_halo_unpack(swe2d_handle *h, int i_buffer, const double *recv_buf_dev);
/* ---- peer-to-peer halo exchange through IPC-mapped device memory: the exchange as two kernels on the handle's stream (no host
 * or RCCL call in the step loop, capturable in a HIP graph).  Same role as swe2d_halo_pack + send/recv + swe2d_halo_unpack.
 * After swe2d_halo_setup:  swe2d_p2p_create (landing zone for n_recv cells x 2 slots per channel; channel 0 = the SWE state,
 * width 3k doubles per cell, channel 1 + t = tracer t, width k)  ->  swe2d_p2p_export (64-byte hipIpcMemHandle_t to hand to
 * the peers through any side channel, e.g. an all-gather; local_base serves peers inside the same process)  ->  every rank
 * swe2d_p2p_open()s the zones of the ranks it sends to  ->  swe2d_p2p_connect: peer i receives send-list cells
 * [send_offset[i], + send_count[i]) at cell offset remote_recv_offset[i] of ITS recv list (remote_n_recv[i] cells long) and
 * knows me as its sender number remote_flag_index[i]; n_from = number of ranks that send to me (their flag indices are
 * 0..n_from-1).  Then, per exchange and channel, on every rank in the same order: swe2d_p2p_push ... swe2d_p2p_wait_unpack.
 * A rank may run at most one exchange ahead of a peer (double-buffered slots); waits are bounded (SWE2D_OPT_P2P_TIMEOUT_MS,
 * default 5 s) and counted in swe2d_p2p_status.timeouts instead of hanging the device. */
#define SWE2D_IPC_HANDLE_BYTES 64
int  swe2d_p2p_create(swe2d_handle *h, int32_t n_channels, const int32_t *widths);
int  swe2d_p2p_export(swe2d_handle *h, void *ipc_handle_out, void **local_base, int32_t *zone_kind);
int  swe2d_p2p_open(swe2d_handle *h, const void *ipc_handle, void **remote_base);
int  swe2d_p2p_connect(swe2d_handle *h, int32_t n_peers, void *const *remote_base, const int32_t *send_offset,
                       const int32_t *send_count, const int32_t *remote_recv_offset, const int32_t *remote_flag_index,
                       const int32_t *remote_n_recv, int32_t n_from);
int  swe2d_p2p_push(swe2d_handle *h, int channel, int i_buffer);
int  swe2d_p2p_wait_unpack(swe2d_handle *h, int channel, int i_buffer);
/* the same for up to four channels (the shallow water state and the tracers at the end of a coupled exchange cycle) in ONE launch each
 * way: the channels' own kernels side by side, every channel with its own epochs and flags - interchangeable with the
 * single-channel calls, exchange by exchange */
int  swe2d_p2p_push_multi(swe2d_handle *h, int n, const int32_t *channels, const int32_t *i_buffers);
int  swe2d_p2p_wait_unpack_multi(swe2d_handle *h, int n, const int32_t *channels, const int32_t *i_buffers);
/* swe2d_p2p_push / swe2d_p2p_wait_unpack on a stream of their own (null: the handle's stream), so that the stage kernels of the
 * interior cells run while the halo travels - PyOP2 overlaps its halo exchange with the core of a par_loop the same way
 * [FD-assumed].  The caller orders the two streams with events; no synchronisation here. */
int  swe2d_set_exchange_stream(swe2d_handle *h, void *hip_stream);
int  swe2d_p2p_status(swe2d_handle *h, int64_t *epochs_sent, int64_t *epochs_received, int32_t *timeouts);
/* ERKGenericShuOsher.solve_stage restricted to cells [cell_begin, cell_end) (may include ghost layers) */
int  swe2d_solve_stage_cells(swe2d_handle *h, int i_stage, int32_t cell_begin, int32_t cell_end);
/* solve_stage(0) on [0, cell_end_0) followed by solve_stage(1) on [0, cell_end_1) (rungekutta.py:930-946; cell_end_1 <= cell_end_0
 * and every cell of the second range has its facet neighbours inside the first: the stage ranges of a partition's exchange cycle, or
 * the whole mesh twice) as ONE launch by overlapped tiles where that kernel covers the handle (swe2d_fused_pair_info), as the two
 * stage launches otherwise.  Either way state buffer 2 holds stage_sol[1] on [0, cell_end_1) afterwards, bit for bit the same;
 * stage_sol[0] is only left behind by the stage launches. */
int  swe2d_solve_stage_pair_cells(swe2d_handle *h, int32_t cell_end_0, int32_t cell_end_1);
/* ForwardEuler (timeintegrator.py:115-165) on a partition: the step from state buffer 0 into buffer 1 on a cell range;
 * after the last range of a step swe2d_swap_state_buffers makes buffer 1 the state (halo pack / unpack take the buffer
 * index).  Not capturable in a replayed graph across an odd number of swaps. */
int  swe2d_forward_euler_cells(swe2d_handle *h, int32_t cell_begin, int32_t cell_end);
int  swe2d_swap_state_buffers(swe2d_handle *h);
/* n_stages (a multiple of 3, at most 384) consecutive solve_stage calls - i.e. n_stages / 3 calls of ERKGenericShuOsher.advance
 * (rungekutta.py:949-952) - in ONE launch without a grid-wide barrier between the stages (csrc/swe2d_flow.h): stage s updates
 * the cells [0, cell_end[s]) (cell_end non-increasing, and every cell of stage s + 1's range has its facet neighbours inside
 * stage s's range: the shrinking ranges of a partition's exchange cycle, or n_owned throughout), a 64-cell block starts stage
 * s + 1 as soon as the trace values it needs from the blocks around it have arrived.  State buffer 0 (swe2d_get_state) ends bit for
 * bit as the swe2d_solve_stage_cells calls it stands for leave it; the intermediate stage solutions (swe2d_get_stage_state 0, 1)
 * are not produced.  swe2d_advance uses it on its own where it applies (SWE2D_OPT_FLOW = 0: never).  Needs every block of the
 * handle resident at once:
 * swe2d_flow_supported returns 0 when the mesh is too large for that (or the configuration is not covered: quadrilaterals,
 * viscosity; wetting-drying IS covered since round 5 - csrc/swe2d_k_flow_wd.hip, SWE2D_OPT_FLOW_WD = 0 leaves it to the stage
 * launches), 1 covered, 2 covered and without source terms;
 * SWE2D_ERR_UNSUPPORTED from swe2d_solve_flow otherwise.  Every wait inside the kernel is bounded (SWE2D_OPT_FLOW_TIMEOUT_MS, default 2 s); a timeout invalidates the
 * state and is reported as SWE2D_ERR_HIP by the next swe2d_synchronize / swe2d_get_state / swe2d_diagnostics
 * (swe2d_flow_status reads the count without failing). */
int  swe2d_solve_flow(swe2d_handle *h, int32_t n_stages, const int32_t *cell_end);
int  swe2d_flow_supported(swe2d_handle *h);
/* The same with the halo exchange of a partition INSIDE the launch, cell by cell: n_cycles exchange cycles (at most 64) of
 * stages_per_cycle stages each (at most 384 stages in all) on the ranges cell_end[0 .. stages_per_cycle) of ONE cycle.  Needs the
 * peer-to-peer halo connected with a LAST channel of width 18 (swe2d_p2p_create: nine 16-byte {value, push number} granules per
 * cell; the channel's epoch flags are not used).  Every cycle but the launch's first starts by receiving what the peers pushed at
 * the end of their previous cycle - every ghost cell's lane waits for the nine granules of ITS cell, so a ghost cell is ready as
 * soon as the one block of the peer that owns it has finished (the first cycle receives too if a push of an earlier launch is
 * still pending: pushes > receives of that channel in swe2d_p2p_status) - and ends by storing this rank's send cells as granules
 * into the peers' zones.  One launch replaces n_cycles x (swe2d_solve_flow + swe2d_p2p_push + swe2d_p2p_wait_unpack); the LAST
 * cycle's push is received by the next launch of this kind or by swe2d_flow_unpack_pending (needed before other kernels read the
 * ghost cells).  Cells sent to more than two peers are not supported (SWE2D_ERR_UNSUPPORTED).  Same results bit for bit. */
int  swe2d_solve_flow_exchange(swe2d_handle *h, int32_t n_cycles, int32_t stages_per_cycle, const int32_t *cell_end);
/* builds the tables of the former ahead of its first launch (which otherwise does it: allocations and a stream synchronisation,
 * not allowed inside a stream capture); after swe2d_halo_setup and swe2d_flow_set_order */
int  swe2d_flow_prepare_exchange(swe2d_handle *h);
int  swe2d_flow_unpack_pending(swe2d_handle *h);
/* The kernel's 64-cell blocks are consecutive cells of a FLOW ORDER (default: the numbering of swe2d_mesh).  A partition whose
 * ghost layers are appended to the numbering layer by layer (what the stage ranges need) passes an order - a permutation of
 * the cell ids - in which every ghost cell sits next to the cells it touches: blocks then stay compact patches and few facets
 * cross block rims.  Results do not depend on the order.  Synchronises the stream; not inside a stream capture. */
int  swe2d_flow_set_order(swe2d_handle *h, const int32_t *cells_in_flow_order);
int  swe2d_flow_status(swe2d_handle *h, int32_t *timeouts);
int  swe2d_debug_flow_poke(swe2d_handle *h, int32_t block, int32_t delta);      /* test hook: skews one block's stage counter */
/* test hook of the -DSWE_FLOW_DELAY build (an adversary for the granule protocol, csrc/swe2d_flow.h): one block sleeps at chosen
 * points of its stage loop; where = 1 before polling | 2 before publishing | 4 before an in-launch receive | 8 before an
 * in-launch push.  SWE2D_ERR_UNSUPPORTED in the product build. */
int  swe2d_debug_flow_delay(swe2d_handle *h, int32_t block, int32_t where, int32_t microseconds, int32_t every_nth_stage);
/* test hook of the -DSWE_FLOW_TEAR build (the adversary of the granules' check word, csrc/swe2d_flow.h): block `block` (-2: every
 * block, -1: off) makes the granule stores of every n-th publish in two halves, the half with the new tag first and the value
 * `microseconds` later; across_ranks: also the pushes into the peers' landing zones.  block = -3 asks whether the build's consumers
 * test the check word (1; 0 in the -DSWE_FLOW_NOCHECK negative control).  SWE2D_ERR_UNSUPPORTED in the product build. */
int  swe2d_debug_flow_tear(swe2d_handle *h, int32_t block, int32_t microseconds, int32_t every_nth_publish, int32_t across_ranks);
/* run on a caller-provided hipStream_t (e.g. torch's current stream) instead of the handle's own */
int  swe2d_set_stream(swe2d_handle *h, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SWE2D_H */
