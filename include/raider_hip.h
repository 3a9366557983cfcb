/* raider_hip.h - C ABI of libraider_hip.so, the MI355X (gfx950) engine behind RAiDER's delay hot path.
 *
 * Drop-in boundary: these are the entry points a RAiDER maintainer's ctypes binding would call in
 * place of the reference's NumPy/scipy/pyproj hot loops and its two native extensions.  Every entry
 * cites the reference interface it replaces (paths relative to the RAiDER repository root).
 * INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  Every function returns an int status
 *     (RDR_OK or a negative RDR_ERR_*); rdr_last_error() gives the message.
 *   - the caller owns every buffer it passes; the library owns device memory behind the opaque
 *     rdr_ctx / rdr_cube handles and never returns an allocation.
 *   - `loc` says where the caller's arrays live: RDR_HOST (NumPy) - the library stages them through
 *     its own device scratch and copies results back; RDR_DEVICE - pointers are HIP device pointers
 *     (e.g. torch tensors), kernels are launched on the ctx stream and NOT synchronised.
 *   - all floating point arrays are float64 unless a dtype argument says otherwise; angles in degrees,
 *     lengths in metres; NaN is the in-band missing value exactly as in the reference.
 *   - one ctx = one device + one stream; a ctx is not thread-safe, distinct ctxs are.
 *   - argument errors never reach the device: a NULL pointer where an array is needed, a negative count, an unknown mode /
 *     dtype / direction is RDR_ERR_INVALID with the reason in rdr_last_error; a count of 0 is an empty batch (RDR_OK, nothing
 *     touched).  A HIP failure the library reports (RDR_ERR_HIP, e.g. out of memory) leaves no error state behind: the
 *     next call starts clean.  rdr_last_error(ctx) is the context's last message, rdr_last_error(NULL) the calling thread's.
 */
#ifndef RAIDER_HIP_H
#define RAIDER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDR_OK 0
#define RDR_ERR_INVALID (-1)    /* bad argument                -> TypeError / ValueError in the shim       */
#define RDR_ERR_HIP (-2)        /* HIP runtime failure         -> RuntimeError                             */
#define RDR_ERR_NODEVICE (-3)   /* no usable GPU               -> RuntimeError (product never falls back)  */
#define RDR_ERR_ALL_NAN (-4)    /* every ray length is NaN     -> ValueError('geo2rdr did not converge...') delay.py:279-280 */
#define RDR_ERR_NO_LEVELS (-5)  /* no model interval contributes: build_ray returns (None,None,None) losreader.py:832-833 */
#define RDR_ERR_NAN_LENGTH (-6) /* some (not all) ray lengths NaN: the reference's nParts (delay.py:283) is undefined there */
#define RDR_ERR_OOM (-7)        /* the device could not hold an allocation (hipErrorOutOfMemory) -> MemoryError; the caller may retry with a smaller batch */

#define RDR_F32 0
#define RDR_F64 1
#define RDR_BYTESWAPPED 0x100 /* or-ed into the dtype of rdr_cube_create: the source fields are in the OTHER byte order than the host
                               * (a NetCDF-3 file is big-endian) - swapped on the device while packing, so a file mapping is uploaded as is */
#define RDR_HOST 0
#define RDR_DEVICE 1

/* ray origin modes */
#define RDR_ORIGIN_GRID 0 /* meshgrid(xpts, ypts) at height ht, row-major (ny,nx)  delay.py:242,262-267 */
#define RDR_ORIGIN_LLH 1  /* per-ray lat[n], lon[n] at height ht                                   */
#define RDR_ORIGIN_XYZ 2  /* per-ray ECEF xyz[n,3] (already at height ht)                          */
/* look-vector modes */
#define RDR_LOS_VEC 0      /* los[n,3] unit ECEF, ground->sensor: what los.getLookVectors returns, delay.py:270 */
#define RDR_LOS_INC_HD 1   /* inc[n], hd[n] degrees -> inc_hd_to_enu + enu2ecef  losreader.py:374-396, utilFcns.py:91-121 */
#define RDR_LOS_INC_HD_SCALAR 2 /* one inc0/hd0 for every ray                                       */
#define RDR_LOS_ZENITH 3   /* getZenithLookVecs, losreader.py:302-316                               */

/* flag bits returned by rdr_ray_prepass / consumed by rdr_ray_march */
#define RDR_FLAG_ANY_NAN 1        /* some ray length is NaN                                          */
#define RDR_FLAG_ANY_FINITE 2     /* some ray length is finite                                       */
#define RDR_FLAG_FIRST_NOT_BELOW 4 /* some ray's first sample is NOT below min(model_zs)  (delay.py:306) */
#define RDR_FLAG_LAST_NOT_ABOVE 8  /* some ray's last sample is NOT above max(model_zs)   (delay.py:310) */
#define RDR_FLAG_DIVERGED 16      /* a level's maximum ray length asks for more than 65536 integration parts (or is not
                                    * finite): the level crossings diverged, e.g. look vectors far from unit length.  The
                                    * slice's outputs are NaN and synchronous calls return RDR_ERR_INVALID.               */

#define RDR_FLAG_BAD_HEIGHT 32    /* per-ray heights: some rays->hts[i] lies below the `ht` the batch's level table was built for:
                                    * the outputs are NaN and synchronous calls return RDR_ERR_INVALID.                        */

#define RDR_FLAG_NAN_OUTPUT 64    /* rdr_raytrace_slices: some delay of the slice is NaN (what delay.py:187 scans the result for) */

typedef struct rdr_ctx rdr_ctx;
typedef struct rdr_cube rdr_cube;

/* Ray batch = one (ny,nx) slice of _build_cube_ray at one height `ht` (delay.py:256-273). */
typedef struct rdr_rays {
    int64_t n;          /* number of rays (ny*nx in GRID mode)                                  */
    int32_t origin_mode;
    int32_t los_mode;
    int64_t nx, ny;     /* GRID mode                                                            */
    const double* xpts; /* GRID: [nx] lon deg                                                   */
    const double* ypts; /* GRID: [ny] lat deg                                                   */
    const double* lat;  /* LLH: [n]                                                             */
    const double* lon;  /* LLH: [n]                                                             */
    const double* xyz;  /* XYZ: [n,3]; with an inc/heading or zenith LOS lat/lon are needed too */
    const double* los;  /* LOS_VEC: [n,3]                                                       */
    const double* inc;  /* LOS_INC_HD: [n]                                                      */
    const double* hd;   /* LOS_INC_HD: [n], or NULL: heading hd0 for every ray                  */
    double inc0, hd0;   /* LOS_INC_HD_SCALAR (hd0 also: LOS_INC_HD with hd == NULL)             */
    int32_t loc;        /* RDR_HOST / RDR_DEVICE for every pointer above                        */
    int32_t _pad;
    const double* hts;  /* NULL: every ray starts at the `ht` of the call (the reference's slice, delay.py:256-273).
                         * [n]: PER-RAY origin heights (a scene on a DEM; BASELINE configs "c3b").  The reference has no
                         * such batch; the rule is its slice algorithm ray by ray: the level tests of
                         * losreader.py:785-808 with the ray's own height (its first contributing interval gets the
                         * 10-iteration crossings and fixes its cos_factor), nParts[k] from the maximum over the rays level k
                         * contributes to (delay.py:283), the all-pixels z-clamp asked about every ray's own first / last
                         * sample.  Equal heights reproduce the slice result bit for bit.  The `ht` argument of the ray
                         * entry points must then be <= min(hts) - normally min(hts): it fixes the batch's level table
                         * (K, the layout of maxlen / nparts); a ray whose no interval contributes gets 0.
                         * GRID / LLH origins are placed at hts[i]; XYZ origins are taken as given.  Not with
                         * rdr_raytrace_slices.                                                     */
} rdr_rays;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int rdr_version(void);
/* sha256[:16] over the sources the library was compiled from (every .h / .hip under raider_amd/csrc + this header; "unknown" when the
 * build recipe did not pass it).  raider_amd._lib.source_hash() computes the same digest from the tree. */
const char* rdr_source_hash(void);
/* device < 0: use HIP's current device.  Fails with RDR_ERR_NODEVICE when there is no GPU. */
int rdr_create(int device, rdr_ctx** out);
void rdr_destroy(rdr_ctx* ctx);
/* message of the last failure on this thread (ctx may be NULL) */
const char* rdr_last_error(rdr_ctx* ctx);
/* Launch on an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  NULL is HIP's default stream;
 * (void*)-1 goes back to the ctx's private stream.  RDR_DEVICE arrays are only ordered against work on THIS stream. */
int rdr_set_stream(rdr_ctx* ctx, void* hip_stream);
/* Tell the context that a stream handed to rdr_set_stream is about to be destroyed: its pending work is waited for if it is still the
 * current stream (the context then goes back to its private stream) and no event is recorded on it ever again (buffers of destroyed
 * cubes are recycled behind events on the streams that used them).  Streams that live as long as the process (torch's) need no call. */
int rdr_forget_stream(rdr_ctx* ctx, void* hip_stream);
int rdr_synchronize(rdr_ctx* ctx);
int rdr_device_info(rdr_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* total_mem);
/* profiling: while on, a HIP event pair brackets every kernel launch on the ctx stream (no sync).
 * rdr_set_profiling(on) also resets the counters.  rdr_profile_get synchronises on the recorded
 * events and returns how many launches of kernel kind `which` (0 ray prepass, 1 ray march, 2 interp,
 * 3 other) were recorded and their summed duration in ms. */
int rdr_set_profiling(rdr_ctx* ctx, int on);
/* Page-locked host memory for result arrays (hipHostMalloc): downloads into it run at the link rate, without first-touch page
 * faults, and asynchronously - rdr_raytrace_slices overlaps them with the kernels of the next slices.  The Python layer keeps a
 * recycling pool of such blocks behind the delay cubes tropo_delay returns (raider_amd/_pinned.py). */
int rdr_host_alloc(int64_t bytes, void** out);
int rdr_host_free(void* p);
/* Ray pass 1 hands each ray's record to pass 2 through an HBM workspace of 232 B per ray (29 doubles: the degree-5 ray
 * polynomials h(u), lat(u), lon(u), the degree-7 level-crossing polynomial, the ray-length scale and the two crossings of
 * the first level; 3.7 GB for 16 M rays).  The few rays the static classification sends to the generic-geodesy kernels
 * (poles, grazing incidence, the +-180 deg meridian) keep origin / look vector / origin frame in the same record and their
 * K+1 level crossings in a compact side buffer sized for 1/32 of the batch (or for the generic-ray count last seen on this
 * ctx); a generic ray that finds it full has its crossings recomputed by pass 2.
 * `bytes` caps records + side buffer (default 48 GiB of the 288 GB, and never more than half of the free memory); larger
 * batches are integrated in chunks.  Env override at rdr_create: RAIDER_HIP_WORKSPACE_BYTES. */
int rdr_set_workspace_limit(rdr_ctx* ctx, int64_t bytes);
/* Give device memory back: the context keeps its scratch (staging slots, the intermediate cube of rdr_point_delays, the ray-record
 * workspace) and up to 4 GiB of value buffers of destroyed cubes for the next call of the same size.  rdr_trim waits for the
 * context's streams, then frees every scratch buffer larger than `keep_bytes` and pooled buffers until at most `keep_bytes` remain
 * (0: everything); *released (may be NULL) = bytes freed.  rdr_point_delays trims its own intermediates above
 * RAIDER_HIP_SCRATCH_KEEP_BYTES (default 4 GiB) before it returns. */
int rdr_trim(rdr_ctx* ctx, int64_t keep_bytes, int64_t* released);
/* Columns of the generic-ray side buffer: >= 0 fixes the capacity (0: always recompute), -1 restores the automatic sizing. */
int rdr_set_side_capacity(rdr_ctx* ctx, int64_t columns);
int rdr_profile_get(rdr_ctx* ctx, int which, int* count, float* total_ms);
/* diagnostics: the shader clock the chip ACTUALLY runs at while the ctx's kernels execute (it is managed by power: fp64-dense kernels
 * run near 2.0 GHz, not at the data sheet's 2.4).  _begin puts one sleeping wave on the ctx's copy stream for `ms` of wall time; it
 * reads the shader-clock counter and the 100 MHz wall counter before and after.  _end waits for it and returns cycles / wall time
 * in GHz.  Launch the kernels to be observed between the two calls. */
int rdr_clock_sample_begin(rdr_ctx* ctx, double ms);
int rdr_clock_sample_end(rdr_ctx* ctx, double* ghz);
/* diagnostics: resources of the light ray kernel a GRID + look-vector batch on `cube` launches (which 0: pass 1 crossings_kernel,
 * 1: pass 2 march_kernel; 2 / 3: their per-ray-height instantiations, rdr_rays.hts), read from the loaded code object (hipFuncGetAttributes): vector registers per lane, static LDS bytes,
 * dynamic LDS bytes of the launch (axis / level tables), scratch bytes per lane, max threads per block.  Any output may be NULL. */
int rdr_ray_kernel_attributes(rdr_ctx* ctx, const rdr_cube* cube, int which, int32_t* vgprs, int32_t* static_lds, int32_t* dynamic_lds,
                              int32_t* scratch, int32_t* max_threads);
/* diagnostics: rays the static classification sent to the generic-geodesy kernels in the last ray pass 1 whose result this ctx read
 * back (rdr_ray_prepass, and rdr_raytrace / rdr_raytrace_slices when they synchronise); -1 for a NULL ctx */
int64_t rdr_generic_ray_count(rdr_ctx* ctx);

/* ---- weather cube ----------------------------------------------------------------------------
 * Replaces getInterpolators (tools/RAiDER/delayFcns.py:23-58): the two fields of one processed weather
 * model (`wet`/`hydro` f32, or `wet_total`/`hydro_total` f64) on a rectilinear (y,x,z) grid, wrapped
 * with scipy-RegularGridInterpolator semantics (linear, bounds_error=False, fill_value=nan).
 * wet/hydro: element [iy,ix,iz] at  base + iy*sy + ix*sx + iz*sz  (ELEMENT strides), so both the
 * file order (z,y,x) (weatherModel.py:685-693) and the transposed (y,x,z) order are accepted without
 * a host-side transpose.  Axes may be ascending or descending (scipy flips; so do we, on device).
 * The device copy interleaves (wet,hydro) per cell, z fastest.  */
int rdr_cube_create(rdr_ctx* ctx, const double* ys, int64_t ny, const double* xs, int64_t nx,
                    const double* zs, int64_t nz, const void* wet, const void* hydro, int dtype,
                    int64_t sy, int64_t sx, int64_t sz, int loc, rdr_cube** out);
void rdr_cube_destroy(rdr_cube* cube);
/* Synchronisation of cube-making entries (round 5).  loc == RDR_HOST: the call returns when the cube is complete (the host arrays may be
 * reused at once).  loc == RDR_DEVICE - cubes made from device arrays, the intermediate delay cubes of rdr_build_cube_to_cube /
 * rdr_raytrace_slices_to_cube - and rdr_cube_blend: NO host synchronisation; the cube is complete in stream order on the context's stream,
 * every entry that reads it from another stream (or another context) first makes that stream wait for the cube's ready event, and the
 * source device arrays must stay valid in stream order, as for any asynchronous kernel.  rdr_set_stream orders the new stream after
 * everything enqueued on the previous one (the context's scratch and flag words are shared by them).
 * Small HOST inputs of any entry (the axes of a cube, AOI axes, height lists; up to 1 MiB each) go up through a ring of page-locked
 * buffers: the caller's array is consumed when the call returns and the host does not wait for work queued earlier on the stream - so the
 * creation of a cube from device arrays, and rdr_build_cube_to_cube / rdr_raytrace_slices_to_cube with host axes, really are asynchronous
 * (round 6; before, the pageable copy of the axes waited for the stream).  Larger host inputs are waited for (one event behind their copy):
 * a host array may be reused as soon as the call that took it returns, whatever memory it lives in.
 * rdr_cube_has_nan: 1 when a NaN was seen among the two source fields while the cube was packed (what delayFcns.py:50-52 scans for on the
 * host: "Weather model contains NaNs!"), 0 otherwise (blended cubes: either source's; a view: its source's), -1 for NULL.  For an
 * asynchronously made cube this call is where the host waits (for that cube's ready event, once). */
int rdr_cube_has_nan(const rdr_cube* cube);
int rdr_cube_shape(const rdr_cube* cube, int64_t* ny, int64_t* nx, int64_t* nz, int* dtype);
/* ascending copies of the axes as the interpolator's `.grid` exposes them (delay.py:239) */
int rdr_cube_axes(const rdr_cube* cube, double* ys, double* xs, double* zs);
/* Projected weather models (HRRR Lambert conformal conic, models/hrrr.py:248-259): tell the cube that its x/y axes are
 * projected metres.  Every entry point that takes GEODETIC query coordinates (rdr_build_cube's xpts/ypts = lon/lat,
 * the ray tracer's samples) then applies geodetic -> model CRS on the device - the pyproj step of delay.py:207-209,253,295.
 * kind RDR_PROJ_LCC, params = {a, es, lat_1, lat_2, lat_0, lon_0, x_0, y_0} (m, -, deg, deg, deg, deg, m, m); es = 0 for
 * HRRR's sphere (a = 6371229).  rdr_interp3 keeps taking points already in cube coordinates (scipy semantics). */
#define RDR_PROJ_LONLAT 0
#define RDR_PROJ_LCC 1
#define RDR_PROJ_STERE 2   /* POLAR stereographic: params = {a, es, lat_0 (+-90), lat_ts (NaN: use k_0), k_0, lon_0, x_0, y_0} (HRRR-AK, models/hrrr.py:22-25) */
int rdr_cube_set_projection(rdr_cube* cube, int kind, const double* params, int nparams);
/* A VIEW of a cube: a second handle on the same device buffers (values, axes, corner-quad copy - nothing is copied or uploaded)
 * that owns only its projection.  The reference builds its pyproj transformers afresh in every call (delay.py:196-216,238-253) and
 * shares nothing between calls; a cube cached per weather-model file and handed to several callers (threads, model CRS arguments)
 * must therefore never be re-projected in place - each call takes a view with the projection IT was given and destroys it afterwards.
 * kind / params as rdr_cube_set_projection; kind -1: the source's own projection.  `ctx` = the context the view is used from (may
 * differ from the source's).  The buffers live until the source AND every view are destroyed, in any order. */
int rdr_cube_view(rdr_ctx* ctx, const rdr_cube* cube, int kind, const double* params, int nparams, rdr_cube** out);
/* transformPoints (delay.py:404-436) for EPSG:4326 -> the cube's CRS: (lat, lon) deg -> (y, x) model coordinates */
int rdr_project_points(rdr_ctx* ctx, const rdr_cube* cube, const double* lat, const double* lon, int64_t n, double* y, double* x, int loc);
/* transformPoints (delay.py:404-436) between EPSG:4326 and a TRANSVERSE-MERCATOR CRS (every UTM zone, EPSG:326xx / 327xx; national
 * TM grids) - the pyproj call behind output grids that are not lon/lat (delay.py:207-209,259-263).  params = {a, es, lat_0, lon_0,
 * k_0, x_0, y_0} (m, -, deg, deg, -, m, m).  direction 0: in = (lat, lon) deg -> out = (y, x) m; 1: in = (y, x) m -> out = (lat, lon) deg. */
int rdr_transform_tm(rdr_ctx* ctx, const double* params, int nparams, int direction, const double* in_a, const double* in_b, int64_t n,
                     double* out_a, double* out_b, int loc);
/* The same between EPSG:4326 and a CONIC model CRS (kind RDR_PROJ_LCC / RDR_PROJ_STERE, params as rdr_cube_set_projection): the
 * pyproj call of transformPoints(lats, lons, hgts, EPSG:4326, hrrr_proj) and back (test/test_delayFcns.py:67-84 round-trips it).
 * direction 0: in = (lat, lon) deg -> out = (y, x) m; 1: in = (y, x) m -> out = (lat, lon) deg. */
int rdr_transform_cone(rdr_ctx* ctx, int kind, const double* params, int nparams, int direction, const double* in_a, const double* in_b,
                       int64_t n, double* out_a, double* out_b, int loc);
/* temporal blend, cli/raider.py:817-819: out = w1*a + w2*b (f32 cubes blend in f32, f64 in f64) */
int rdr_cube_blend(rdr_ctx* ctx, const rdr_cube* a, double w1, const rdr_cube* b, double w2, rdr_cube** out);
/* Azimuth-time-grid temporal interpolation.
 * rdr_inverse_time_weights = get_inverse_weights_for_dates (s1_azimuth_timing.py:326-399): az[n] = per-voxel acquisition
 * times and dates[nd] = model times, all in seconds on one epoch; weights[d*n + i] = normalised inverse-|dt| weight of
 * date d at voxel i, masked to |dt| <= window_s (window_s < 0: min_i |dates[i]-dates[0]|, the inferred model step).
 * RDR_ERR_INVALID: duplicate dates / no date ("No dates provided are within temporal window").
 * rdr_cube_blend_weighted = the combination of cli/raider.py:817-819 with those per-voxel weights: weights[d][nz*ny*nx]
 * in FILE order (z,y,x) (the shape of the reference's time grid), result a new f64 cube (a float64 weight array
 * promotes the f32 fields in the reference too). */
int rdr_inverse_time_weights(rdr_ctx* ctx, const double* az, int64_t n, const double* dates, int32_t nd, double window_s,
                             double regularizer, double* weights, int loc);
int rdr_cube_blend_weighted(rdr_ctx* ctx, const rdr_cube* const* cubes, int32_t nd, const double* weights, int loc, rdr_cube** out);

/* GUNW radian conversion (aria/calcGUNW.py:54-59): out = delay * (-4 pi / wavelength) for both fields;
 * dtype RDR_F32 multiplies in f32 by the f32-rounded factor (NumPy weak-scalar rule), RDR_F64 in f64.  In place allowed. */
int rdr_delays_to_phase(rdr_ctx* ctx, const void* wet, const void* hydro, int64_t n, int dtype, double wavelength,
                        void* wet_out, void* hydro_out, int loc);

/* copy the (blended) fields back, (y,x,z) C-order, dtype of the cube */
int rdr_cube_read(rdr_ctx* ctx, const rdr_cube* cube, void* wet, void* hydro);

/* ---- zenith / projected path -----------------------------------------------------------------
 * scipy RegularGridInterpolator.__call__ on both fields (delay.py:214,120-121): pts[n,3] = (y,x,z). */
/* Either of wet / hydro may be NULL: that field is then neither stored nor downloaded (a caller that evaluates the two interpolators one
 * after the other - `for intp in interpolators: intp(pts)`, delay.py:213-214 - moves 8 B per point and call instead of 16). */
int rdr_interp3(rdr_ctx* ctx, const rdr_cube* cube, const double* pts, int64_t n, double* wet,
                double* hydro, int loc);
/* The second stage of tropo_delay's point branch (delay.py:110-128) in one call: the gather of rdr_interp3 on the intermediate delay
 * cube and, for a projected line of sight, the division of Conventional.__call__ (losreader.py:130-133) before the values leave the
 * device - 24-32 B per point up, 16 B down, nothing in between.
 * Points: three arrays y[n], x[n], z[n] (what aoi.readLL() / readZ() hand out: no packed copy on the host), or y = packed pts[n,3]
 * with x = z = NULL.  proj_mode 0: no projection; 1: proj[n] = incidence angles (deg), delay / cosd(inc) (inc_hd_to_enu(...)[..., -1],
 * losreader.py:374-396); 2: one incidence inc0 for every point (proj unused); 3: proj[n] = the divisor itself (the cosine of the look
 * angle state_to_los returns for an orbit file, losreader.py:122-128).  Either output may be NULL. */
int rdr_interp3_project(rdr_ctx* ctx, const rdr_cube* cube, const double* y, const double* x, const double* z, int64_t n, int proj_mode,
                        const double* proj, double inc0, double* wet, double* hydro, int loc);
/* Large random point sets on a cube beyond the caches (BASELINE configs[4]: 5 M stations on a 1000 x 1000 x 50 cube): rdr_interp3
 * then reads four 128 B lines per point for 16 B each.  The cube can carry a second, cell-column-major copy ("corner quads",
 * 5.3 x its bytes for f32, 8 x for f64) from which a point's eight corners are ONE line - same values, same arithmetic, 3.3 x less
 * HBM traffic.  mode 1: build it now; mode 0: free it.  Without this call rdr_interp3 builds it by itself from the second call with
 * >= 262144 points on a cube beyond 192 MB (the Infinity Cache holds smaller ones) - or at the first, when the point set is large enough that the build pays for itself within
 * that call (n x 175 B > the copy's bytes, from the measured rates) - if it fits a quarter of the free memory (env
 * RAIDER_HIP_POINT_INDEX=0 never, =1 at the first such call, =2 second call only).  rdr_cube_point_index_bytes: bytes the copy holds now (0: none). */
int rdr_cube_point_index(rdr_ctx* ctx, rdr_cube* cube, int mode);
int64_t rdr_cube_point_index_bytes(const rdr_cube* cube);
/* _build_cube (delay.py:196-216) for model_crs == pts_crs: out[(iz*ny+iy)*nx+ix] = f(ypts[iy],xpts[ix],zpts[iz]) */
int rdr_build_cube(rdr_ctx* ctx, const rdr_cube* cube, const double* xpts, int64_t nx, const double* ypts,
                   int64_t ny, const double* zpts, int64_t nz, double* wet, double* hydro, int loc);
/* 1 / 0: the result of the last rdr_build_cube call with HOST arrays on this ctx holds / does not hold a NaN - the scan the caller
 * runs over the result (delay.py:187) done on the device before the download; -1: unknown (no such call yet, or device arrays). */
int rdr_last_nan_output(rdr_ctx* ctx);
/* _build_cube whose result STAYS on the device as a new float64 cube with axes (ypts, xpts, zpts) - the intermediate delay cube of
 * tropo_delay's point branch (delay.py:96-121: _get_delays_on_cube -> writeResultsToXarray -> getInterpolators(ds, 'ztd')), ready
 * for rdr_interp3_project.  Nothing crosses PCIe but the three axes.  rdr_cube_has_nan(*out) answers the scan of delay.py:187
 * ("There are missing delay values").  Needs two nodes per axis (scipy's grid rule), nz <= 512. */
int rdr_build_cube_to_cube(rdr_ctx* ctx, const rdr_cube* cube, const double* xpts, int64_t nx, const double* ypts, int64_t ny,
                           const double* zpts, int64_t nz, int loc, rdr_cube** out);
/* rdr_interp3 on the two-epoch temporal blend w1 * a + w2 * b (cli/raider.py:817-819) WITHOUT making the blended cube: the blend is
 * applied at the eight corners of every point, in the cubes' own dtype with blend_kernel's arithmetic - the same bits as rdr_cube_blend
 * followed by rdr_interp3.  It reads eight lines per point instead of four and none of the blend's 24 B per cell: the better deal for a
 * point set below ~5 % of the cube's cells - the station block of one rank of a multi-GPU job (BASELINE configs[4]), where a replicated
 * blend is what stops the job from scaling.  Points as rdr_interp3_project (three arrays, or y = packed pts[n,3] with x = z = NULL). */
int rdr_interp3_blend(rdr_ctx* ctx, const rdr_cube* a, double w1, const rdr_cube* b, double w2, const double* y, const double* x, const double* z,
                      int64_t n, double* wet, double* hydro, int loc);
/* The same query when the point set is LARGE against the cube (one GPU holding all 5 M stations of BASELINE configs[4]): the blend is made
 * as a cube after all - 24 B per cell, as rdr_cube_blend - but into the context's scratch and in the layout the gather reads best
 * (neighbouring x columns paired: 3 cache lines per random point on average instead of 4), then gathered in the same call.  The paired
 * cube never becomes an rdr_cube, so nothing else can be handed that layout.  Same arithmetic as rdr_cube_blend + rdr_interp3
 * (cli/raider.py:817-819, delay.py:116-121): the same bits.  Points / outputs as rdr_interp3_blend. */
int rdr_interp3_blend_cube(rdr_ctx* ctx, const rdr_cube* a, double w1, const rdr_cube* b, double w2, const double* y, const double* x, const double* z,
                           int64_t n, double* wet, double* hydro, int loc);
/* tropo_delay's point branch for a zenith / projected line of sight (delay.py:96-128) in ONE call, host arrays in, host arrays out:
 * rdr_build_cube_to_cube on the output grid (xpts[nx], ypts[ny], zpts[nz]) + rdr_interp3_project at the query points, with the
 * intermediate cube in the context's scratch (no allocation per call) and the upload of the points running under its build.  Same
 * kernels and arithmetic as the two separate entries: the same bits.  Points / projection / outputs as rdr_interp3_project;
 * *cube_has_nan (may be NULL): the intermediate cube holds a NaN - the scan of delay.py:187 ("There are missing delay values"). */
int rdr_point_delays(rdr_ctx* ctx, const rdr_cube* cube, const double* xpts, int64_t nx, const double* ypts, int64_t ny, const double* zpts,
                     int64_t nz, const double* y, const double* x, const double* z, int64_t n, int proj_mode, const double* proj, double inc0,
                     double* wet, double* hydro, int32_t* cube_has_nan);
/* Conventional.__call__ tail (losreader.py:130-133): delay / cosd(inc) in place, inc[n] in degrees (what inc_hd_to_enu(...)[..., -1]
 * holds for an incidence raster).  The reference projects wet and hydro in two calls: either pointer may be NULL. */
int rdr_project_cosinc(rdr_ctx* ctx, double* wet, double* hydro, const double* inc, int64_t n, int loc);
/* The same with the divisor given (losreader.py:130-131: LOS_enu from an orbit file is cos(look angle), same shape as the delays). */
int rdr_project_divide(rdr_ctx* ctx, double* wet, double* hydro, const double* divisor, int64_t n, int loc);

/* ---- ray-traced path --------------------------------------------------------------------------
 * rdr_ray_levels: the slice-uniform part of build_ray (losreader.py:785-808).  lo/hi/kz need room for
 * nz-1 entries; kz[k] = index of the model interval. Returns RDR_ERR_NO_LEVELS when K==0. */
int rdr_ray_levels(const rdr_cube* cube, double ht, double zref, int32_t* K, double* lo, double* hi, int32_t* kz);
/* Pass 1 (build_ray, losreader.py:772-835, fused - nothing materialised): per-level max ray length over
 * the batch (what delay.py:283 reduces) and the RDR_FLAG_* bits.  maxlen (K doubles) and flags are HOST
 * outputs (this call synchronises).  Multi-GPU callers all-reduce MAX(maxlen) / OR(flags) across ranks. */
int rdr_ray_prepass(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, double ht, double zref,
                    double* maxlen, int32_t* flags);
/* nParts = ceil(maxlen/max_seg)+1 (delay.py:283) */
int rdr_nparts(const double* maxlen, int32_t K, double max_seg, int32_t* nparts);
/* Device-resident variant of the prepass / march pair for multi-GPU slabs: `partition` is a DEVICE buffer of
 * K+4 doubles - the per-level maxima of the ray length followed by the four RDR_FLAG_* bits as 0.0 / 1.0 - so that the
 * caller can run ONE element-wise MAX all-reduce (RCCL) on it between the two calls and no host round trip happens at all.
 * Rays, outputs and the partition must live on the device; both calls are asynchronous on the ctx stream.  The march
 * derives nParts = ceil(max/max_seg)+1 (delay.py:283) on the device; error conditions (all-NaN, diverged) are not
 * reported here (the outputs are NaN) - use rdr_ray_prepass when they must be. */
int rdr_ray_prepass_device(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, double ht, double zref, double* partition);
int rdr_ray_march_device(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, double ht, double zref, double max_seg,
                         const double* partition, double* wet, double* hydro);

/* Pass 2 (delay.py:285-323 + build_ray recomputed in registers): trapezoid integral of both fields along
 * each ray with the GIVEN partition nparts[K] (host array) and clamp decision `flags`.
 * wet/hydro: [n] outputs (overwritten, not accumulated), location = rays->loc. */
int rdr_ray_march(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, double ht, double zref,
                  const int32_t* nparts, int32_t flags, double* wet, double* hydro);
/* One slice of _build_cube_ray (delay.py:256-323): prepass -> nParts on device -> march, no host
 * round trip.  nparts_out (host, K entries, may be NULL) and flags_out (may be NULL) are filled after
 * the launch (this forces a synchronisation; pass NULL for a fully asynchronous call with RDR_DEVICE).
 * Returns RDR_ERR_ALL_NAN / RDR_ERR_NAN_LENGTH only when it synchronised. */
int rdr_raytrace(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, double ht, double zref,
                 double max_seg, double* wet, double* hydro, int32_t* nparts_out, int32_t* flags_out);

/* The height loop of _build_cube_ray (delay.py:256-323) in one call: `nslices` slices at heights hts[] (host array) of the SAME
 * origins - GRID or LLH - each integrated exactly as rdr_raytrace would (its own level table, per-level slice maxima, nParts and
 * z-clamp decision), but in one pass-1 / pass-2 launch pair, so that production-sized jobs (20 heights x 1e4-1e5 rays,
 * aria/prepFromGUNW.py:173,180) fill the GPU.  los_per_slice != 0: rays->los / inc / hd hold nslices consecutive blocks of
 * rays->n entries (look vectors that depend on the target height, losreader.py:219-255); 0: one block serves every slice.
 * wet / hydro: [nslices][n].  Host outputs, filled after a synchronisation when given: K_out[nslices] contributing model
 * intervals per slice (0: build_ray -> None, the slice's delays are 0), nparts_out[nslices][ld] (ld >= nz-1),
 * flags_out[nslices] RDR_FLAG_* bits - the caller raises what delay.py:276-283 raises, slice by slice. */
int rdr_raytrace_slices(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, const double* hts, int32_t nslices,
                        int32_t los_per_slice, double zref, double max_seg, double* wet, double* hydro, int32_t* K_out,
                        int32_t* nparts_out, int32_t ld, int32_t* flags_out);

/* rdr_raytrace_slices whose delays STAY on the device as a new float64 cube with axes (rays->ypts, rays->xpts, hts): the intermediate
 * cube of tropo_delay's point branch for a ray-traced line of sight (delay.py:96-121).  GRID batches, >= 2 nodes per axis, strictly
 * monotonic hts; the partition outputs are as for rdr_raytrace_slices. */
int rdr_raytrace_slices_to_cube(rdr_ctx* ctx, const rdr_cube* cube, const rdr_rays* rays, const double* hts, int32_t nslices,
                                int32_t los_per_slice, double zref, double max_seg, int32_t* K_out, int32_t* nparts_out, int32_t ld,
                                int32_t* flags_out, rdr_cube** out);

/* Materialising variants for API parity on small inputs:
 * getTopOfAtmosphere (losreader.py:706-733): factor==NULL -> 10 iterations with factor 1, else 3 */
int rdr_top_of_atmosphere(rdr_ctx* ctx, const double* xyz, const double* los, int64_t n, double toaheight,
                          const double* factor, double* pos, int loc);
/* build_ray (losreader.py:772-835): ray_lengths[K,n], low[K,n,3], high[K,n,3] */
int rdr_build_ray(rdr_ctx* ctx, const double* model_zs, int64_t nz, double ht, const double* xyz,
                  const double* los, int64_t n, double zref, int32_t* K, double* lengths, double* low,
                  double* high, int loc);

/* ---- geodesy helpers (pyproj call sites utilFcns.py:77-88; LOS helpers losreader.py:302-316,374-396) */
int rdr_lla2ecef(rdr_ctx* ctx, const double* lat, const double* lon, const double* h, int64_t n, double* xyz, int loc);
int rdr_ecef2lla(rdr_ctx* ctx, const double* xyz, int64_t n, double* lon, double* lat, double* h, int loc);
/* look vectors for a ray batch (any los_mode) -> los[n,3] */
int rdr_look_vectors(rdr_ctx* ctx, const rdr_rays* rays, double ht, double* los);

/* Front end of the cube producer for ECMWF hybrid model levels (ERA-5, HRES): utilFcns.calcgeoh (:781-859) +
 * utilFcns.geo_to_ht (:378-410) + the re-ordering of models/ecmwf.py:92-110.  z_surf (surface geopotential) and lnsp: [ny*nx] f32;
 * t, q: [nlev, ny, nx] f32 with level 1 = model top (file order); lats[ny] f32; a, b: [nlev+1] hybrid coefficients (host).
 * p_out, zs_out: [ny, nx, nlev] f64, BOTTOM level first - the layout rdr_cubes_from_model_levels takes.  Arithmetic is float64
 * on the float32 inputs: the reference's float32 evaluation of these formulas is ill-conditioned (heights off by up to 2.4 m,
 * platform dependent in the last bit of logf); DESIGN.md 6.5. */
int rdr_ecmwf_model_levels(rdr_ctx* ctx, const float* z_surf, const float* lnsp, const float* t, const float* q, const float* lats,
                           const double* a, const double* b, int32_t nlev, int64_t ny, int64_t nx, double R_d,
                           double* p_out, double* zs_out, int loc);

/* Cube producer (models/weatherModel.py:235-262: _find_e -> _uniform_in_z -> _checkForNans -> wet/hydro refractivity ->
 * _adjust_grid -> _getZTD): model-level columns zs3/p/t/hum [ny, nx, nlev] (heights ascending along the last axis,
 * humidity_type 0 = specific humidity q, 1 = relative humidity %) are resampled to the uniform levels new_z[nz] and
 * turned into the two device cubes the delay kernels read, without a NetCDF round trip:
 *   *pointwise = (wet, hydro) refractivity, f32;   *total = (wet_total, hydro_total) zenith delays, f64.
 * When zmin < new_z[0] an extra bottom level at zmin is added (the reference's padLower).  t_out/p_out/e_out (f32,
 * [ny, nx, nz_out], all three or none) return the resampled state for parity checks.  k1,k2,k3: models/ecmwf.py:26-28. */
int rdr_cubes_from_model_levels(rdr_ctx* ctx, const double* ys, int64_t ny, const double* xs, int64_t nx, const double* zs3,
                                const double* p, const double* t, const double* hum, int humidity_type, int64_t nlev,
                                const double* new_z, int64_t nz, double k1, double k2, double k3, double zmin, int loc,
                                rdr_cube** pointwise, rdr_cube** total, float* t_out, float* p_out, float* e_out);

/* Look vectors from orbit state vectors: replaces the per-pixel isce3 geo2rdr + Orbit.interpolate loop of
 * Raytracing.getLookVectors (losreader.py:219-255; zero-Doppler, threshold 1e-7, maxiter 30 at the call site).
 * sv_t[nsv] seconds (strictly increasing, HOST), sv_pos/sv_vel [nsv,3] ECEF (HOST); xyz[n,3] targets, los[n,3] unit
 * vectors target->sensor (NaN where the solve fails or leaves the orbit span); aztime / srange optional [n].
 * isce3 itself is not available to this build: parity with it is UNPINNED (the algorithm is restated from its
 * published description). */
int rdr_orbit_look_vectors(rdr_ctx* ctx, const double* sv_t, const double* sv_pos, const double* sv_vel, int64_t nsv,
                           const double* xyz, int64_t n, double threshold, int maxiter, double* los, double* aztime,
                           double* srange, int loc);

/* ---- the reference's two native extensions ----------------------------------------------------
 * RAiDER.interpolate.interpolate (tools/bindings/interpolate/src/module.cpp:26-294, interpolate.cpp):
 * N-D linear interpolation (ndim <= 8 on device), C-order values, interp_points[n,ndim].
 * has_fill: queries with upper-bound index outside [1, N-1] (incl. ON the last node) get fill_value;
 * else indices are clamped (linear extrapolation).  interpolate.h:23-74 */
int rdr_interp_nd(rdr_ctx* ctx, int32_t ndim, const double* const* axes, const int64_t* axis_len,
                  const double* values, const double* interp_points, int64_t n, int has_fill,
                  double fill_value, double* out, int loc);
/* RAiDER.interpolate.interpolate_along_axis (module.cpp:296-493, interpolate.cpp:260-332) on arrays made
 * contiguous with the axis LAST: points/values [ncol, m], interp_points/out [ncol, mq]. */
int rdr_interp_along_axis(rdr_ctx* ctx, const double* points, const double* values, int64_t ncol, int64_t m,
                          const double* interp_points, int64_t mq, int has_fill, double fill_value,
                          double* out, int loc);
/* RAiDER.makePoints.makePoints{0,1,2,3}D (tools/bindings/utils/makePoints.pyx:15-148):
 * out[r,3,npts] = sp[r,c] + (k*step)*slv[r,c]; npts = rdr_make_points_count(max_len, step) */
int64_t rdr_make_points_count(double max_len, double step);
int rdr_make_points(rdr_ctx* ctx, double max_len, const double* sp, const double* slv, int64_t nrays,
                    double step, double* out, int loc);

#ifdef __cplusplus
}
#endif
#endif /* RAIDER_HIP_H */
