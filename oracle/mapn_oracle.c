/*
 * mapn_oracle.c -- CPU restatement of the n-body compute step.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped library (libmapn.so) does not link,
 * load or call anything in oracle/.
 *
 * PARITY UNPINNED: the reference tree (/root/reference) holds no tests, golden vectors or
 * fixtures for this path, has no CPU simulation path, and cannot be compiled here (Compute.cpp:31
 * needs <ppl.h>, D3D12, DirectXMath; the step itself exists only as HLSL).  This restatement
 * follows the reference text line by line instead; the known answers it is checked against
 * (tests/test_oracle_kat.py) are derived from that text by hand.
 *
 * What is restated, with the reference lines each function follows (paths relative to
 * /root/reference/Particles/):
 *   pair term                nBodyGravityCS.hlsl:44-57   (bodyBodyInteraction)
 *   central-well force       nBodyGravityCS.hlsl:92-101  (CSMain as shipped)
 *   integrator + outputs     nBodyGravityCS.hlsl:103-108
 *   constants                nBodyGravityCS.hlsl:37-38, Compute.cpp:542-546, defines.h:37,39,42
 *   active-count rounding    Compute.cpp:1041            (groups of 64, no in-shader bound)
 *   initial state            Compute.cpp:820-844 (two halves), :711-749 (LCG variant), :599-609
 *   LCG known answers        Compute.cpp:599-609 (fast_rand), :622-661 (rand_sse)
 *
 * Besides the reference-order step, three DIAGNOSTIC variants of the all-pairs sum exist so that a
 * device-vs-oracle difference can be attributed instead of asserted:
 *   FP64_ACC       pair term in fp32 exactly as the HLSL, accumulated in double, rounded once;
 *   ORDER_MATCHED  the summation ORDER and operation FUSION of the device kernel
 *                  (multi-adapter-particles_amd/csrc/mapn_kernels.hip, pair_term2 / finish): the
 *                  64-body tiles of the j-range cut into S chunks, each chunk summed over ascending
 *                  j with fused multiply-adds, chunk sums added in ascending order (waves inside a
 *                  workgroup, then rows), mass applied after the sum, fma in the integrator.  What
 *                  is left between this mode and the device is v_rsq_f32 vs 1/sqrtf alone;
 *   step_all_pairs_f64   the whole step in double on double state ("truth" of the discrete map).
 * None of them restates the reference; the parity statements are made against the REFERENCE mode.
 *
 * Arithmetic contract: every operation is a separately rounded IEEE-754 binary32 operation in
 * the order the HLSL source writes it (build with -ffp-contract=off, no -ffast-math).  The
 * all-pairs sum runs over j = 0..N-1 ascending into ONE accumulator per component.  The code is
 * vectorised across i (16 bodies per block), never across j, so the result is bit-identical to
 * a scalar loop on any x86-64 CPU.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define MAPN_ORACLE_BLOCK 64 /* defines.h:37 BLOCK_SIZE */
#define IB 16                /* i-bodies per vector block */

typedef struct {
    float mass;     /* nBodyGravityCS.hlsl:38 g_fParticleMass (default 70000) */
    float soft2;    /* nBodyGravityCS.hlsl:37 softeningSquared (default 25)  */
    float dt;       /* Compute.cpp:545 paramf[0] = 0.1f */
    float damping;  /* Compute.cpp:546 paramf[1] = 1.0f */
} mapn_oracle_params;

void mapn_oracle_default_params(mapn_oracle_params *p)
{
    p->mass = 70000.0f;
    p->soft2 = 25.0f;
    p->dt = 0.1f;
    p->damping = 1.0f;
}

/* Compute.cpp:542-546 -- the 32-byte constant block uploaded at init (K8). */
void mapn_oracle_cbuffer(uint32_t num_particles, uint32_t out_param[4], float out_paramf[4])
{
    out_param[0] = num_particles;
    out_param[1] = (uint32_t)(int)ceilf((float)num_particles / (float)MAPN_ORACLE_BLOCK);
    out_param[2] = 0;
    out_param[3] = 0;
    out_paramf[0] = 0.1f;
    out_paramf[1] = 1.0f;
    out_paramf[2] = 0.0f;
    out_paramf[3] = 0.0f;
}

/* Compute.cpp:1041 -- Dispatch(ceil(numActive/64)) x 64 threads, D3D drops out-of-range writes. */
uint32_t mapn_oracle_active_bodies(int num_active, uint32_t num_particles)
{
    if (num_active <= 0) return 0;
    uint64_t groups = ((uint64_t)num_active + MAPN_ORACLE_BLOCK - 1) / MAPN_ORACLE_BLOCK;
    uint64_t n = groups * MAPN_ORACLE_BLOCK;
    return (uint32_t)(n < num_particles ? n : num_particles);
}

/* nBodyGravityCS.hlsl:44-57 -- one softened pair term; bj.w / bi.w ignored; no i != j test. */
void mapn_oracle_pair_term(float ai[3], const float bj[4], const float bi[4], float mass,
                           int particles, float soft2)
{
    float rx = bj[0] - bi[0];                       /* :46 */
    float ry = bj[1] - bi[1];
    float rz = bj[2] - bi[2];
    float d = rx * rx + ry * ry;                    /* :48 dot(r,r) */
    d = d + rz * rz;
    d = d + soft2;                                  /* :49 */
    float inv = 1.0f / sqrtf(d);                    /* :51 */
    float inv3 = inv * inv * inv;                   /* :52 */
    float s = mass * inv3 * (float)particles;       /* :54 */
    ai[0] = ai[0] + rx * s;                         /* :56 */
    ai[1] = ai[1] + ry * s;
    ai[2] = ai[2] + rz * s;
}

/* nBodyGravityCS.hlsl:103-108 -- kick, damp, drift; w = |accel|. */
static inline void integrate(const float *pos, const float *vel, float ax, float ay, float az,
                             const mapn_oracle_params *p, float *npos, float *nvel)
{
    float vx = vel[0] + ax * p->dt;                 /* :103 */
    float vy = vel[1] + ay * p->dt;
    float vz = vel[2] + az * p->dt;
    vx = vx * p->damping;                           /* :104 */
    vy = vy * p->damping;
    vz = vz * p->damping;
    float px = pos[0] + vx * p->dt;                 /* :105 */
    float py = pos[1] + vy * p->dt;
    float pz = pos[2] + vz * p->dt;
    float l = ax * ax + ay * ay;
    l = l + az * az;
    npos[0] = px;                                   /* :107 float4(pos.xyz, length(accel)) */
    npos[1] = py;
    npos[2] = pz;
    npos[3] = sqrtf(l);
    nvel[0] = vx;                                   /* :108 */
    nvel[1] = vy;
    nvel[2] = vz;
}

/*
 * CSMain exactly as shipped (nBodyGravityCS.hlsl:86-109): one gravity well at the origin.
 * Advances bodies [first, first+count) from (old_pos, old_vel) into (new_pos, new_vel).
 * pos: float4 stride 16 (Render.h:85-88); vel: packed float3 stride 12 (Compute.h:66-69).
 */
void mapn_oracle_step_central_well(const float *old_pos, const float *old_vel, float *new_pos,
                                   float *new_vel, uint32_t first, uint32_t count,
                                   const mapn_oracle_params *p)
{
    for (uint32_t i = first; i < first + count; i++) {
        const float *pos = old_pos + 4 * (size_t)i;
        float rx = pos[0], ry = pos[1], rz = pos[2];    /* :92 */
        float d = rx * rx + ry * ry;                    /* :94 */
        d = d + rz * rz;
        d = d + p->soft2;                               /* :95 */
        float inv = -1.0f / sqrtf(d);                   /* :97 */
        float inv3 = inv * inv * inv;                   /* :98 */
        float s = p->mass * inv3;                       /* :99 */
        integrate(pos, old_vel + 3 * (size_t)i, rx * s, ry * s, rz * s, p,
                  new_pos + 4 * (size_t)i, new_vel + 3 * (size_t)i);
    }
}

typedef struct {
    const float *old_pos, *old_vel;
    float *new_pos, *new_vel;
    uint32_t n_total, first, count;
    const mapn_oracle_params *p;
    uint32_t tid, nthreads;
} ap_job;

/* bodies [b0, b0+nb) (nb <= IB) against j = 0..n_total-1 ascending, one accumulator each.
 * Function multi-versioning: the AVX-512 / AVX2 / baseline clones execute the same separately
 * rounded IEEE operations per lane, so every clone returns the same bits; the widest one the
 * host supports is picked at load time (the .so is built in one container and timed in another). */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void all_pairs_block(const ap_job *J, uint32_t b0, uint32_t nb)
{
    float xi[IB], yi[IB], zi[IB], ax[IB], ay[IB], az[IB];
    const float mass = J->p->mass, soft2 = J->p->soft2;
    for (uint32_t k = 0; k < IB; k++) {
        uint32_t i = b0 + (k < nb ? k : 0);
        xi[k] = J->old_pos[4 * (size_t)i + 0];
        yi[k] = J->old_pos[4 * (size_t)i + 1];
        zi[k] = J->old_pos[4 * (size_t)i + 2];
        ax[k] = ay[k] = az[k] = 0.0f;
    }
    const float *pj = J->old_pos;
    for (uint32_t j = 0; j < J->n_total; j++, pj += 4) {
        const float xj = pj[0], yj = pj[1], zj = pj[2];
#pragma GCC ivdep
        for (int k = 0; k < IB; k++) {             /* the pair term, hlsl:44-57, particles = 1 */
            float rx = xj - xi[k];
            float ry = yj - yi[k];
            float rz = zj - zi[k];
            float d = rx * rx + ry * ry;
            d = d + rz * rz;
            d = d + soft2;
            float inv = 1.0f / sqrtf(d);
            float inv3 = inv * inv * inv;
            float s = mass * inv3 * 1.0f;
            ax[k] = ax[k] + rx * s;
            ay[k] = ay[k] + ry * s;
            az[k] = az[k] + rz * s;
        }
    }
    for (uint32_t k = 0; k < nb; k++) {
        uint32_t i = b0 + k;
        integrate(J->old_pos + 4 * (size_t)i, J->old_vel + 3 * (size_t)i, ax[k], ay[k], az[k],
                  J->p, J->new_pos + 4 * (size_t)i, J->new_vel + 3 * (size_t)i);
    }
}

static void *all_pairs_worker(void *arg)
{
    const ap_job *J = (const ap_job *)arg;
    uint32_t nblocks = (J->count + IB - 1) / IB;
    for (uint32_t b = J->tid; b < nblocks; b += J->nthreads) {
        uint32_t b0 = J->first + b * IB;
        uint32_t nb = J->first + J->count - b0;
        all_pairs_block(J, b0, nb < IB ? nb : IB);
    }
    return NULL;
}

int mapn_oracle_hardware_threads(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

/*
 * All-pairs step: CSMain's integrator (hlsl:103-108) with lines :92-101 replaced by
 *   accel = sum_{j=0}^{n_total-1} bodyBodyInteraction(accel, oldPosition[j], pos, mass, 1)
 * Advances bodies [first, first+count); j always runs over all n_total old positions (the
 * reference uploads param[0] = N, Compute.cpp:543).  Parallel over i like the reference's only
 * host loop (concurrency::parallel_for over i, Compute.cpp:684).  threads <= 0: all cores.
 * The result does not depend on the thread count.
 */
int mapn_oracle_step_all_pairs(const float *old_pos, const float *old_vel, float *new_pos,
                               float *new_vel, uint32_t n_total, uint32_t first, uint32_t count,
                               const mapn_oracle_params *p, int threads)
{
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    uint32_t nblocks = (count + IB - 1) / IB;
    if ((uint32_t)threads > nblocks) threads = nblocks ? (int)nblocks : 1;
    ap_job *jobs = (ap_job *)calloc((size_t)threads, sizeof(ap_job));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < threads; t++) {
        jobs[t] = (ap_job){old_pos, old_vel, new_pos, new_vel, n_total, first, count, p,
                           (uint32_t)t, (uint32_t)threads};
        if (t > 0 && pthread_create(&th[t], NULL, all_pairs_worker, &jobs[t]) != 0) {
            all_pairs_worker(&jobs[t]);
            th[t] = 0;
        }
    }
    all_pairs_worker(&jobs[0]);
    for (int t = 1; t < threads; t++)
        if (th[t]) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* diagnostic variants of the all-pairs sum (see header)                                       */

enum { MAPN_ORACLE_SUM_REFERENCE = 0, MAPN_ORACLE_SUM_FP64_ACC = 1, MAPN_ORACLE_SUM_ORDER_MATCHED = 3 };

typedef struct {
    int mode;            /* MAPN_ORACLE_SUM_* */
    uint32_t waves, sb;  /* ORDER_MATCHED: the device plan, S = waves * sb chunks (mapn_kernel_stats: block_x / 64, grid_y) */
} mapn_oracle_sum_spec;

typedef struct {
    ap_job base;
    mapn_oracle_sum_spec spec;
} apx_job;

/* nBodyGravityCS.hlsl:103-108 with the device kernel's fusion (integrate_store in mapn_kernels.hip):
 * v = fma(a, dt, v) * damping; x = fma(v, dt, x); w = sqrt(fma(az, az, fma(ay, ay, ax * ax))) */
static inline void integrate_fused(const float *pos, const float *vel, float ax, float ay, float az,
                                   const mapn_oracle_params *p, float *npos, float *nvel)
{
    float vx = __builtin_fmaf(ax, p->dt, vel[0]) * p->damping;
    float vy = __builtin_fmaf(ay, p->dt, vel[1]) * p->damping;
    float vz = __builtin_fmaf(az, p->dt, vel[2]) * p->damping;
    npos[0] = __builtin_fmaf(vx, p->dt, pos[0]);
    npos[1] = __builtin_fmaf(vy, p->dt, pos[1]);
    npos[2] = __builtin_fmaf(vz, p->dt, pos[2]);
    npos[3] = sqrtf(__builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)));
    nvel[0] = vx; nvel[1] = vy; nvel[2] = vz;
}

/* FP64_ACC: the fp32 pair term of hlsl:44-57 op by op, summed in double over ascending j */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void all_pairs_block_acc64(const ap_job *J, uint32_t b0, uint32_t nb)
{
    float xi[IB], yi[IB], zi[IB];
    double ax[IB], ay[IB], az[IB];
    const float mass = J->p->mass, soft2 = J->p->soft2;
    for (uint32_t k = 0; k < IB; k++) {
        uint32_t i = b0 + (k < nb ? k : 0);
        xi[k] = J->old_pos[4 * (size_t)i + 0];
        yi[k] = J->old_pos[4 * (size_t)i + 1];
        zi[k] = J->old_pos[4 * (size_t)i + 2];
        ax[k] = ay[k] = az[k] = 0.0;
    }
    const float *pj = J->old_pos;
    for (uint32_t j = 0; j < J->n_total; j++, pj += 4) {
        const float xj = pj[0], yj = pj[1], zj = pj[2];
        /* (two loops: the fp32 pair term at the full vector width, then the widening adds.  This leg costs 2.4 - 3 x the reference-order
         *  one on the GPU box's host whichever way it is written -- one mixed loop, this form, vector-typed accumulators were all measured:
         *  six conversions and six double adds per sixteen pairs beside one sqrt and one divide; same bits in every form) */
        float tx[IB], ty[IB], tz[IB];
#pragma GCC ivdep
        for (int k = 0; k < IB; k++) {
            float rx = xj - xi[k];
            float ry = yj - yi[k];
            float rz = zj - zi[k];
            float d = rx * rx + ry * ry;
            d = d + rz * rz;
            d = d + soft2;
            float inv = 1.0f / sqrtf(d);
            float inv3 = inv * inv * inv;
            float s = mass * inv3 * 1.0f;
            tx[k] = rx * s; ty[k] = ry * s; tz[k] = rz * s;  /* the fp32 products of hlsl:56 */
        }
#pragma GCC ivdep
        for (int k = 0; k < IB; k++) {
            ax[k] = ax[k] + (double)tx[k];
            ay[k] = ay[k] + (double)ty[k];
            az[k] = az[k] + (double)tz[k];
        }
    }
    for (uint32_t k = 0; k < nb; k++) {
        uint32_t i = b0 + k;
        integrate(J->old_pos + 4 * (size_t)i, J->old_vel + 3 * (size_t)i, (float)ax[k], (float)ay[k], (float)az[k],
                  J->p, J->new_pos + 4 * (size_t)i, J->new_vel + 3 * (size_t)i);
    }
}

/* ORDER_MATCHED: chunk c of S owns tiles [c*base + min(c, rem), ... + base + (c < rem)) of the
 * ceil(N/64) 64-body tiles (chunk_tiles in mapn_kernels.hip); inside a chunk
 *   d = fma(dz,dz, fma(dy,dy, fma(dx,dx, soft2))); inv = 1/sqrt(d); inv3 = (inv*inv)*inv;
 *   a = fma(dx, inv3, a)                                   (pair_term2)
 * chunk sums are added waves-ascending into a workgroup sum, workgroup sums rows-ascending into
 * the total (finish<> / reduce_integrate_kernel), the total is multiplied by the mass. */
__attribute__((target_clones("avx512f", "fma", "default")))
static void all_pairs_block_matched(const apx_job *X, uint32_t b0, uint32_t nb)
{
    const ap_job *J = &X->base;
    float xi[IB], yi[IB], zi[IB], tx[IB], ty[IB], tz[IB], wx[IB], wy[IB], wz[IB], cx[IB], cy[IB], cz[IB];
    const float soft2 = J->p->soft2;
    for (uint32_t k = 0; k < IB; k++) {
        uint32_t i = b0 + (k < nb ? k : 0);
        xi[k] = J->old_pos[4 * (size_t)i + 0];
        yi[k] = J->old_pos[4 * (size_t)i + 1];
        zi[k] = J->old_pos[4 * (size_t)i + 2];
        tx[k] = ty[k] = tz[k] = 0.0f;
    }
    const uint32_t waves = X->spec.waves, sb = X->spec.sb, S = waves * sb;
    const uint32_t tiles = (J->n_total + 63u) / 64u, base = tiles / S, rem = tiles % S;
    for (uint32_t row = 0; row < sb; row++) {
        for (int k = 0; k < IB; k++) wx[k] = wy[k] = wz[k] = 0.0f;
        for (uint32_t w = 0; w < waves; w++) {
            const uint32_t c = row * waves + w;
            const uint32_t t0 = c * base + (c < rem ? c : rem), t1 = t0 + base + (c < rem ? 1u : 0u);
            uint32_t j0 = t0 * 64u, j1 = t1 * 64u;
            if (j1 > J->n_total) j1 = J->n_total;
            for (int k = 0; k < IB; k++) cx[k] = cy[k] = cz[k] = 0.0f;
            const float *pj = J->old_pos + 4 * (size_t)j0;
            for (uint32_t j = j0; j < j1; j++, pj += 4) {
                const float xj = pj[0], yj = pj[1], zj = pj[2];
#pragma GCC ivdep
                for (int k = 0; k < IB; k++) {
                    float dx = xj - xi[k], dy = yj - yi[k], dz = zj - zi[k];
                    float d = __builtin_fmaf(dx, dx, soft2);
                    d = __builtin_fmaf(dy, dy, d);
                    d = __builtin_fmaf(dz, dz, d);
                    float inv = 1.0f / sqrtf(d);
                    float inv3 = inv * inv * inv;
                    cx[k] = __builtin_fmaf(dx, inv3, cx[k]);
                    cy[k] = __builtin_fmaf(dy, inv3, cy[k]);
                    cz[k] = __builtin_fmaf(dz, inv3, cz[k]);
                }
            }
            for (int k = 0; k < IB; k++) { wx[k] = wx[k] + cx[k]; wy[k] = wy[k] + cy[k]; wz[k] = wz[k] + cz[k]; }
        }
        for (int k = 0; k < IB; k++) { tx[k] = tx[k] + wx[k]; ty[k] = ty[k] + wy[k]; tz[k] = tz[k] + wz[k]; }
    }
    for (uint32_t k = 0; k < nb; k++) {
        uint32_t i = b0 + k;
        integrate_fused(J->old_pos + 4 * (size_t)i, J->old_vel + 3 * (size_t)i, tx[k] * J->p->mass, ty[k] * J->p->mass,
                        tz[k] * J->p->mass, J->p, J->new_pos + 4 * (size_t)i, J->new_vel + 3 * (size_t)i);
    }
}

static void *all_pairs_worker_ex(void *arg)
{
    const apx_job *X = (const apx_job *)arg;
    const ap_job *J = &X->base;
    uint32_t nblocks = (J->count + IB - 1) / IB;
    for (uint32_t b = J->tid; b < nblocks; b += J->nthreads) {
        uint32_t b0 = J->first + b * IB;
        uint32_t nb = J->first + J->count - b0;
        if (nb > IB) nb = IB;
        if (X->spec.mode == MAPN_ORACLE_SUM_FP64_ACC) all_pairs_block_acc64(J, b0, nb);
        else if (X->spec.mode == MAPN_ORACLE_SUM_ORDER_MATCHED) all_pairs_block_matched(X, b0, nb);
        else all_pairs_block(J, b0, nb);
    }
    return NULL;
}

/* mapn_oracle_step_all_pairs with a selectable summation variant (spec == NULL: reference order) */
int mapn_oracle_step_all_pairs_ex(const float *old_pos, const float *old_vel, float *new_pos,
                                  float *new_vel, uint32_t n_total, uint32_t first, uint32_t count,
                                  const mapn_oracle_params *p, int threads, const mapn_oracle_sum_spec *spec)
{
    mapn_oracle_sum_spec sp = {MAPN_ORACLE_SUM_REFERENCE, 1, 1};
    if (spec) sp = *spec;
    if (sp.mode != MAPN_ORACLE_SUM_REFERENCE && sp.mode != MAPN_ORACLE_SUM_FP64_ACC && sp.mode != MAPN_ORACLE_SUM_ORDER_MATCHED) return -2;
    if (sp.mode == MAPN_ORACLE_SUM_ORDER_MATCHED && (sp.waves == 0 || sp.sb == 0)) return -2;
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    uint32_t nblocks = (count + IB - 1) / IB;
    if ((uint32_t)threads > nblocks) threads = nblocks ? (int)nblocks : 1;
    apx_job *jobs = (apx_job *)calloc((size_t)threads, sizeof(apx_job));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < threads; t++) {
        jobs[t].base = (ap_job){old_pos, old_vel, new_pos, new_vel, n_total, first, count, p, (uint32_t)t, (uint32_t)threads};
        jobs[t].spec = sp;
        if (t > 0 && pthread_create(&th[t], NULL, all_pairs_worker_ex, &jobs[t]) != 0) {
            all_pairs_worker_ex(&jobs[t]);
            th[t] = 0;
        }
    }
    all_pairs_worker_ex(&jobs[0]);
    for (int t = 1; t < threads; t++)
        if (th[t]) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return 0;
}

/* The whole all-pairs step in double on DOUBLE state (pos4: x,y,z,w; vel3): the discrete map
 * itself, free of fp32 rounding -- the yardstick both fp32 paths are measured against. */
typedef struct {
    const double *old_pos, *old_vel;
    double *new_pos, *new_vel;
    uint32_t n_total, first, count;
    double mass, soft2, dt, damping;
    uint32_t tid, nthreads;
} ap64_job;

#define IB64 8
__attribute__((target_clones("avx512f", "avx2", "default")))
static void all_pairs_block_f64(const ap64_job *J, uint32_t b0, uint32_t nb)
{
    double xi[IB64], yi[IB64], zi[IB64], ax[IB64], ay[IB64], az[IB64];
    for (uint32_t k = 0; k < IB64; k++) {
        uint32_t i = b0 + (k < nb ? k : 0);
        xi[k] = J->old_pos[4 * (size_t)i + 0];
        yi[k] = J->old_pos[4 * (size_t)i + 1];
        zi[k] = J->old_pos[4 * (size_t)i + 2];
        ax[k] = ay[k] = az[k] = 0.0;
    }
    const double *pj = J->old_pos;
    const double soft2 = J->soft2;
    for (uint32_t j = 0; j < J->n_total; j++, pj += 4) {
        const double xj = pj[0], yj = pj[1], zj = pj[2];
#pragma GCC ivdep
        for (int k = 0; k < IB64; k++) {
            double rx = xj - xi[k], ry = yj - yi[k], rz = zj - zi[k];
            double d = rx * rx + ry * ry + rz * rz + soft2;
            double inv = 1.0 / sqrt(d);
            double s = inv * inv * inv;
            ax[k] += rx * s; ay[k] += ry * s; az[k] += rz * s;
        }
    }
    for (uint32_t k = 0; k < nb; k++) {
        uint32_t i = b0 + k;
        const double a0 = ax[k] * J->mass, a1 = ay[k] * J->mass, a2 = az[k] * J->mass;
        const double *v = J->old_vel + 3 * (size_t)i, *x = J->old_pos + 4 * (size_t)i;
        double vx = (v[0] + a0 * J->dt) * J->damping, vy = (v[1] + a1 * J->dt) * J->damping, vz = (v[2] + a2 * J->dt) * J->damping;
        double *nx = J->new_pos + 4 * (size_t)i, *nv = J->new_vel + 3 * (size_t)i;
        nx[0] = x[0] + vx * J->dt; nx[1] = x[1] + vy * J->dt; nx[2] = x[2] + vz * J->dt;
        nx[3] = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
        nv[0] = vx; nv[1] = vy; nv[2] = vz;
    }
}

static void *all_pairs_worker_f64(void *arg)
{
    const ap64_job *J = (const ap64_job *)arg;
    uint32_t nblocks = (J->count + IB64 - 1) / IB64;
    for (uint32_t b = J->tid; b < nblocks; b += J->nthreads) {
        uint32_t b0 = J->first + b * IB64;
        uint32_t nb = J->first + J->count - b0;
        all_pairs_block_f64(J, b0, nb < IB64 ? nb : IB64);
    }
    return NULL;
}

int mapn_oracle_step_all_pairs_f64(const double *old_pos, const double *old_vel, double *new_pos, double *new_vel,
                                   uint32_t n_total, uint32_t first, uint32_t count, const mapn_oracle_params *p, int threads)
{
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    uint32_t nblocks = (count + IB64 - 1) / IB64;
    if ((uint32_t)threads > nblocks) threads = nblocks ? (int)nblocks : 1;
    ap64_job *jobs = (ap64_job *)calloc((size_t)threads, sizeof(ap64_job));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < threads; t++) {
        jobs[t] = (ap64_job){old_pos, old_vel, new_pos, new_vel, n_total, first, count,
                             (double)p->mass, (double)p->soft2, (double)p->dt, (double)p->damping, (uint32_t)t, (uint32_t)threads};
        if (t > 0 && pthread_create(&th[t], NULL, all_pairs_worker_f64, &jobs[t]) != 0) {
            all_pairs_worker_f64(&jobs[t]);
            th[t] = 0;
        }
    }
    all_pairs_worker_f64(&jobs[0]);
    for (int t = 1; t < threads; t++)
        if (th[t]) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return 0;
}

/* Per-body accelerations only (no integration): the quantity the 1-step tables compare. */
int mapn_oracle_accel_all_pairs(const float *pos, float *out_acc3, uint32_t n_total,
                                uint32_t first, uint32_t count, float mass, float soft2)
{
    for (uint32_t i = first; i < first + count; i++) {
        float a[3] = {0.0f, 0.0f, 0.0f};
        for (uint32_t j = 0; j < n_total; j++)
            mapn_oracle_pair_term(a, pos + 4 * (size_t)j, pos + 4 * (size_t)i, mass, 1, soft2);
        memcpy(out_acc3 + 3 * (size_t)(i - first), a, sizeof a);
    }
    return 0;
}

/*
 * One Compute::Simulate call on host arrays (Compute.cpp:1009-1055): reads buffer 1-idx, writes
 * buffer idx for bodies [0, active), leaves the rest of buffer idx untouched, flips idx
 * (Compute.cpp:1022,1034-1035,1003; nBodyGravityCS.hlsl:77-81).  mode 0 = all pairs,
 * 1 = central well.  pos[2], vel[2] are the two ping-pong buffers.  Returns the new index.
 */
uint32_t mapn_oracle_simulate(float *pos0, float *pos1, float *vel0, float *vel1,
                              uint32_t buffer_index, uint32_t num_particles, int num_active,
                              int mode, const mapn_oracle_params *p, int threads)
{
    float *pos[2] = {pos0, pos1}, *vel[2] = {vel0, vel1};
    uint32_t w = buffer_index, r = 1 - buffer_index;
    uint32_t active = mapn_oracle_active_bodies(num_active, num_particles);
    if (active) {
        if (mode == 1)
            mapn_oracle_step_central_well(pos[r], vel[r], pos[w], vel[w], 0, active, p);
        else
            mapn_oracle_step_all_pairs(pos[r], vel[r], pos[w], vel[w], num_particles, 0, active, p,
                                       threads);
    }
    return 1 - buffer_index;                        /* Compute.cpp:1003 */
}

/* ------------------------------------------------------------------------------------------ */
/* LCGs of the initial-condition generator                                                     */

/* Compute.cpp:605-609 fast_rand: MSVC rand() LCG.  K5. */
int mapn_oracle_fast_rand(uint32_t *state)
{
    *state = 214013u * *state + 2531011u;
    return (int)((*state >> 16) & 0x7FFF);
}

/*
 * Compute.cpp:622-661 rand_sse restated without SSE: four independent 32-bit LCG lanes.
 * _mm_set_epi32(seed, seed+1, seed, seed+1) puts seed+1 in lanes 0 and 2, seed in lanes 1 and 3
 * (:619); lane multipliers {214013,17405,214013,69069}, adders {2531011,10395331,13737667,1};
 * the 64-bit products are masked back to 32 bits (:648-649) and the result is
 * (state >> 16) & 0x7FFF per lane (:655-657).  K6.
 */
void mapn_oracle_srand_sse(uint32_t state[4], uint32_t seed)
{
    state[0] = seed + 1; state[1] = seed; state[2] = seed + 1; state[3] = seed;
}

void mapn_oracle_rand_sse(uint32_t state[4], int out[4])
{
    static const uint32_t mult[4] = {214013u, 17405u, 214013u, 69069u};
    static const uint32_t gadd[4] = {2531011u, 10395331u, 13737667u, 1u};
    for (int l = 0; l < 4; l++) {
        state[l] = state[l] * mult[l] + gadd[l];
        out[l] = (int)(((int32_t)state[l] >> 16) & 0x7FFF);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Seeded initial state                                                                        */

static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

/* MT19937 (the engine behind std::mt19937, Compute.cpp:680), restated for the USE_ORIG variant */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* std::uniform_real_distribution<float>(-1, 1)(gen) as libstdc++ evaluates it:
 * (b - a) * generate_canonical<float, 24>(gen) + a, one 32-bit draw per value. */
static float mt_uniform(mt19937_t *g)
{
    float c = (float)mt_next(g) / 4294967296.0f;
    if (c >= 1.0f) c = nextafterf(1.0f, 0.0f);
    return 2.0f * c + -1.0f;
}

void mapn_oracle_mt_uniform(uint32_t seed, uint32_t count, float *out)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (uint32_t k = 0; k < count; k++) out[k] = mt_uniform(&g);
}

/*
 * Deterministic replacement for LoadParticles (Compute.cpp:667-812).  The reference seeds
 * mt19937 from random_device and shares it unsynchronised across threads (:678-684), so it is
 * not reproducible; this generator keeps its distribution and seeds the randomness source PER
 * BODY, so the state is independent of thread count and generation order:
 *     body_seed(i) = fmix32(seed * 0x9E3779B9 + i + 1)         (i = global body index)
 * variant 0: the file's own LCG (fast_rand, :599-609, scaled as in :721-725 with MSVC
 *            RAND_MAX = 32767), loop of :723-736;
 * variant 1: rand_sse's four LCG lanes (:619-661), x/y/z from lanes 0/1/2, do-while of :765-777;
 * variant 2: mt19937 + uniform_real_distribution<float>(-1,1) (:679-681), loop of :690-694.
 * XMVector3NormalizeEst (:703-704, an rsqrtps approximation) is replaced by an exact normalize.
 * Bodies [0, N/2) surround (+0.75*spread, 0, 0), [N/2, 2*(N/2)) surround (-0.75*spread, 0, 0)
 * (:831-844).  pos.w = 0.
 */
static void draw_delta(int variant, uint32_t body_seed, float d[3])
{
    const float k_scale = (1.0f / 32767.0f) * 2.0f;             /* :721 */
    if (variant == 1) {
        uint32_t st[4];
        int r[4];
        mapn_oracle_srand_sse(st, body_seed);
        d[0] = d[1] = d[2] = 0.0f;
        for (;;) {
            mapn_oracle_rand_sse(st, r);
            for (int c = 0; c < 3; c++) d[c] = d[c] + ((float)r[c] * k_scale - 1.0f);
            float l = d[0] * d[0] + d[1] * d[1];
            l = l + d[2] * d[2];
            if (!(l < 10.0f)) return;
        }
    }
    if (variant == 2) {
        mt19937_t g;
        mt_seed(&g, body_seed);
        for (int c = 0; c < 3; c++) d[c] = mt_uniform(&g);
        for (;;) {
            float l = d[0] * d[0] + d[1] * d[1];
            l = l + d[2] * d[2];
            if (!(l < 10.0f)) return;
            for (int c = 0; c < 3; c++) d[c] = d[c] + mt_uniform(&g);
        }
    }
    uint32_t st = body_seed;
    for (int c = 0; c < 3; c++) d[c] = (float)mapn_oracle_fast_rand(&st) * k_scale - 1.0f;   /* :723-725 */
    for (;;) {
        float l = d[0] * d[0] + d[1] * d[1];
        l = l + d[2] * d[2];
        if (!(l < 10.0f)) return;                                                   /* :728 */
        for (int c = 0; c < 3; c++) d[c] = d[c] + ((float)mapn_oracle_fast_rand(&st) * k_scale - 1.0f);
    }
}

void mapn_oracle_initial_state_ex(int variant, uint32_t seed, uint32_t n, float spread, float speed,
                                  float *pos4, float *vel3)
{
    const uint32_t half = n / 2;
    const float center_spread = spread * 0.750f;                /* :831 */
    memset(pos4, 0, (size_t)n * 16);
    memset(vel3, 0, (size_t)n * 12);
    for (uint32_t i = 0; i < 2 * half; i++) {
        float d[3];
        draw_delta(variant, fmix32(seed * 0x9E3779B9u + i + 1u), d);
        const float cx = i < half ? center_spread : -center_spread;
        float l = d[0] * d[0] + d[1] * d[1];
        l = l + d[2] * d[2];
        float len = sqrtf(l);
        float px = cx + d[0] / len * spread;                    /* :738-742 */
        float py = 0.0f + d[1] / len * spread;
        float pz = 0.0f + d[2] / len * spread;
        pos4[4 * (size_t)i + 0] = px;
        pos4[4 * (size_t)i + 1] = py;
        pos4[4 * (size_t)i + 2] = pz;
        pos4[4 * (size_t)i + 3] = 0.0f;
        l = px * px + py * py;
        l = l + pz * pz;
        len = sqrtf(l);
        float ux = px / len, uy = py / len, uz = pz / len;              /* :746 direction */
        float qx = 1.0f - ux, qy = 1.0f - uy, qz = 1.0f - uz;           /* :747 */
        l = qx * qx + qy * qy;
        l = l + qz * qz;
        len = sqrtf(l);
        qx = qx / len; qy = qy / len; qz = qz / len;                    /* perp */
        vel3[3 * (size_t)i + 0] = (uy * qz - uz * qy) * speed;          /* :748 cross(dir, perp) */
        vel3[3 * (size_t)i + 1] = (uz * qx - ux * qz) * speed;
        vel3[3 * (size_t)i + 2] = (ux * qy - uy * qx) * speed;
    }
}

void mapn_oracle_initial_state(uint32_t seed, uint32_t n, float spread, float speed, float *pos4,
                               float *vel3)
{
    mapn_oracle_initial_state_ex(0, seed, n, spread, speed, pos4, vel3);
}

/* =================================================================================================
 * ORDER_MATCHED_SYM -- the summation order and operation fusion of the device's SYMMETRIC kernel
 * (multi-adapter-particles_amd/csrc/mapn_sym.hip), the kernel the bench runs by default.  Diagnostic
 * like ORDER_MATCHED above: it does not restate the reference (whose pair term, nBodyGravityCS.hlsl:44-57,
 * is evaluated here once per UNORDERED pair and fed to both bodies); what is left between this mode and
 * the device is v_rsq_f32 vs 1/sqrtf alone.
 *
 * The launch plan is DATA handed in by the test (include/mapn.h: mapn_get_sym_plan /
 * mapn_sym_plan_describe): windows[k] = {g0, g1, meetings of a class-0 block, of a class-1 block};
 * per window bounds[sets][parts * waves + 1] and split[sets][max_meetings] (set = class, or class + 2 * (block mod 8) when the
 * device's parts are XCD-weighted).  Restated, in the device's order:
 *   * blocks of 1024 bodies (the last padded with stand-ins at 3e18 that exert and feel nothing); group g of
 *     block a: 0 = a itself (one-sided), 1..D = partner a + g, D + 1 = the half-ring partner (class 0 only: the runner of the pair, sym_runs_half);
 *     meeting m of a window = group g0 + m / 16, 64-body J-block m % 16 of the partner;
 *   * wave v = part * waves + w runs the linear steps [bounds[v], bounds[v + 1]) (step 64 m + k); lane l owns the
 *     16 bodies a * 1024 + c * 64 + l (c = 0..15) with ONE fma chain each over all of the wave's steps; at step k of
 *     a piece that began with rotation k0 lane l meets J-body (l + k0 + k) % 64:
 *         d = fma(dz,dz, fma(dy,dy, fma(dx,dx, soft2))); inv3 = (inv*inv)*inv; a_i = fma(dx, inv3, a_i)
 *     and the travelling body collects r = fma(-dx, inv3, r) in TWO chains (even c / odd c), visiting c ascending;
 *     a piece's reaction is the sum of the two chains;
 *   * a-row of (block, part) = ((0 + wave 0) + wave 1) + ...; a cut meeting inside one workgroup = first steps + last
 *     steps; cut between two workgroups: first steps in the meeting's row, last steps in the head row;
 *   * per body and window: running sum so far, a-rows in ascending part order, then per group ascending the
 *     meeting's row and its head row (zeros where there is none, padded to batches of eight like the device);
 *     after the last window: total * mass, fused integrator.
 * ================================================================================================= */
typedef struct {
    uint32_t nb, groups, windows, parts, waves, brows, max_meetings, table_stride;
    uint32_t sets;   /* table sets per window: 2 (one per class) or 16 (class + 2 * (block mod 8): the device's XCD-weighted parts) */
} mapn_oracle_sym_shape;

/* which block of a half-ring pair (p, p + nb / 2) runs its meetings: the pairs alternate (csrc/mapn_kernels.h sym_runs_half) */
static inline int sym_runs_half(uint32_t a, uint32_t half)
{
    if (!half) return 0;
    const int low = a < half;
    const uint32_t p = low ? a : a - half;
    return ((p & 1u) == 0u) == low;
}

#define SYM_IB 1024u
#define SYM_JPI 16u
#define SYM_NONE 0xffffffffu

typedef struct {
    const float *old_pos, *old_vel;
    float *new_pos, *new_vel;
    uint32_t n;
    const mapn_oracle_params *p;
    const mapn_oracle_sym_shape *sh;
    const uint32_t *win;        /* this window: g0, g1, m0, m1 */
    const uint32_t *tab;        /* this window's tables */
    float *arow, *brow, *brow1; /* [nb][parts][3][1024], [nb*16][brows][3][64], [nb][parts][3][64] */
    float *acc;                 /* [3][nb*1024] running sum between windows */
    int first_window, last_window;
    uint32_t tid, nthreads;
    uint32_t a0, shard_nbl;     /* ORDER_MATCHED_SHARDED: this rank's first block and block count (0: the whole job in one launch) */
} sym_job;

static inline void sym_body(const sym_job *J, uint32_t i, float *x, float *y, float *z)
{
    if (i < J->n) { *x = J->old_pos[4 * (size_t)i]; *y = J->old_pos[4 * (size_t)i + 1]; *z = J->old_pos[4 * (size_t)i + 2]; }
    else { *x = 3.0e18f; *y = 3.0e18f; *z = 3.0e18f; }
}

/* one workgroup (block a, part s): all its waves, then the combination */
__attribute__((target_clones("avx512f", "fma", "default")))
static void sym_workgroup(const sym_job *J, uint32_t a, uint32_t s, float *scratch)
{
    const mapn_oracle_sym_shape *sh = J->sh;
    const uint32_t nb = sh->nb, D = (nb - 1u) / 2u, half = (nb & 1u) ? 0u : nb / 2u, W = sh->waves;
    const uint32_t g0 = J->win[0];
    const uint32_t cls = sym_runs_half(a, half) ? 0u : 1u;
    const uint32_t la = a - J->a0;                                       /* the block within its launch (a rank's blocks when sharded) */
    const uint32_t set = cls + (sh->sets > 2u ? 2u * (la & 7u) : 0u);
    const uint32_t *bounds = J->tab + set * (sh->parts * W + 1u);
    const float soft2 = J->p->soft2;
    float *xi = scratch, *yi = xi + SYM_IB, *zi = yi + SYM_IB;          /* the I-block */
    float *accw = zi + SYM_IB;                                           /* [W][3][1024] */
    float *edge = accw + (size_t)W * 3u * SYM_IB;                        /* [2][W][3][64] */
    for (uint32_t e = 0; e < SYM_IB; e++) sym_body(J, a * SYM_IB + e, &xi[e], &yi[e], &zi[e]);
    for (uint32_t w = 0; w < W; w++) {
        float *ax = accw + (size_t)w * 3u * SYM_IB, *ay = ax + SYM_IB, *az = ay + SYM_IB;
        for (uint32_t e = 0; e < SYM_IB; e++) ax[e] = ay[e] = az[e] = 0.0f;
        const uint32_t t0 = bounds[s * W + w], t1 = bounds[s * W + w + 1u];
        for (uint32_t t = t0; t < t1;) {
            const uint32_t k0 = t & 63u, steps = (64u - k0 < t1 - t) ? 64u - k0 : t1 - t, m = t >> 6;
            const uint32_t g = g0 + m / SYM_JPI, d = g <= D ? g : half;
            const uint32_t ap = (a + d) % nb, jb = ap * SYM_JPI + m % SYM_JPI;
            float xj[64], yj[64], zj[64], rxe[64], rye[64], rze[64], rxo[64], ryo[64], rzo[64];
            for (uint32_t l = 0; l < 64; l++) {
                sym_body(J, jb * 64u + ((l + k0) & 63u), &xj[l], &yj[l], &zj[l]);
                rxe[l] = rye[l] = rze[l] = rxo[l] = ryo[l] = rzo[l] = 0.0f;
            }
            for (uint32_t k = 0; k < steps; k++) {
                for (uint32_t c = 0; c < 16; c++) {
                    const float *cx = xi + c * 64u, *cy = yi + c * 64u, *cz = zi + c * 64u;
                    float *bx = ax + c * 64u, *by = ay + c * 64u, *bz = az + c * 64u;
                    float *rx = (c & 1u) ? rxo : rxe, *ry = (c & 1u) ? ryo : rye, *rz = (c & 1u) ? rzo : rze;
#pragma GCC ivdep
                    for (int l = 0; l < 64; l++) {
                        const float dx = xj[l] - cx[l], dy = yj[l] - cy[l], dz = zj[l] - cz[l];
                        float dd = __builtin_fmaf(dx, dx, soft2);
                        dd = __builtin_fmaf(dy, dy, dd);
                        dd = __builtin_fmaf(dz, dz, dd);
                        const float inv = 1.0f / sqrtf(dd);
                        const float inv3 = inv * inv * inv;
                        bx[l] = __builtin_fmaf(dx, inv3, bx[l]);
                        by[l] = __builtin_fmaf(dy, inv3, by[l]);
                        bz[l] = __builtin_fmaf(dz, inv3, bz[l]);
                        if (d != 0u) {
                            rx[l] = __builtin_fmaf(-dx, inv3, rx[l]);
                            ry[l] = __builtin_fmaf(-dy, inv3, ry[l]);
                            rz[l] = __builtin_fmaf(-dz, inv3, rz[l]);
                        }
                    }
                }
                /* everything that travels moves one lane on: lane l takes what lane l + 1 held */
                const float fx = xj[0], fy = yj[0], fz = zj[0], a0 = rxe[0], a1 = rye[0], a2 = rze[0], b0 = rxo[0], b1 = ryo[0], b2 = rzo[0];
                for (int l = 0; l < 63; l++) {
                    xj[l] = xj[l + 1]; yj[l] = yj[l + 1]; zj[l] = zj[l + 1];
                    rxe[l] = rxe[l + 1]; rye[l] = rye[l + 1]; rze[l] = rze[l + 1];
                    rxo[l] = rxo[l + 1]; ryo[l] = ryo[l + 1]; rzo[l] = rzo[l + 1];
                }
                xj[63] = fx; yj[63] = fy; zj[63] = fz; rxe[63] = a0; rye[63] = a1; rze[63] = a2; rxo[63] = b0; ryo[63] = b1; rzo[63] = b2;
            }
            t += steps;
            if (d == 0u) continue;
            /* where the piece's reactions go (force_sym_kernel) */
            /* (sharded: one row per (J-block, local I-block) -- the exchange adds them up per destination rank) */
            const size_t row = J->shard_nbl ? (size_t)jb * J->shard_nbl + la : sh->brows ? ((size_t)jb * sh->brows + (g - (g0 ? g0 : 1u))) : 0;
            float *r0 = J->brow + row * 192u, *h0 = J->brow1 + ((size_t)la * sh->parts + s) * 192u;
            for (uint32_t l = 0; l < 64; l++) {
                const float qx = rxe[l] + rxo[l], qy = rye[l] + ryo[l], qz = rze[l] + rzo[l];
                if (steps == 64u) { r0[l] = qx; r0[64 + l] = qy; r0[128 + l] = qz; }
                else if (k0 != 0u) {                                     /* the meeting's last steps: bodies are home */
                    float *dst = w == 0u ? h0 : edge + ((size_t)(1u * W + w) * 3u) * 64u;
                    dst[l] = qx; dst[64 + l] = qy; dst[128 + l] = qz;
                } else {                                                 /* its first steps: lane l holds body (l + steps) % 64 */
                    const uint32_t home = (l + steps) & 63u;
                    float *dst = w == W - 1u ? r0 : edge + ((size_t)(0u * W + w) * 3u) * 64u;
                    dst[home] = qx; dst[64 + home] = qy; dst[128 + home] = qz;
                }
            }
        }
    }
    /* the workgroup's row: waves in ascending order, starting from zero */
    float *row = J->arow + ((size_t)la * sh->parts + s) * 3u * SYM_IB;
    for (uint32_t e = 0; e < SYM_IB; e++) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (uint32_t w = 0; w < W; w++) {
            const float *ax = accw + (size_t)w * 3u * SYM_IB;
            sx = sx + ax[e]; sy = sy + ax[SYM_IB + e]; sz = sz + ax[2u * SYM_IB + e];
        }
        row[e] = sx; row[SYM_IB + e] = sy; row[2u * SYM_IB + e] = sz;
    }
    /* a symmetric meeting cut between wave w - 1 and wave w: first steps + last steps */
    for (uint32_t w = 1; w < W; w++) {
        const uint32_t t0 = bounds[s * W + w], t1 = bounds[s * W + w + 1u];
        if ((t0 & 63u) == 0u || t0 >= t1) continue;
        const uint32_t m = t0 >> 6, g = g0 + m / SYM_JPI, d = g <= D ? g : half;
        if (d == 0u) continue;
        const uint32_t jb = ((a + d) % nb) * SYM_JPI + m % SYM_JPI;
        float *r0 = J->brow + (J->shard_nbl ? (size_t)jb * J->shard_nbl + la : (size_t)jb * sh->brows + (g - (g0 ? g0 : 1u))) * 192u;
        const float *e0 = edge + ((size_t)(0u * W + w - 1u) * 3u) * 64u, *e1 = edge + ((size_t)(1u * W + w) * 3u) * 64u;
        for (uint32_t l = 0; l < 192; l++) r0[l] = e0[l] + e1[l];
    }
}

static void *sym_force_worker(void *arg)
{
    const sym_job *J = (const sym_job *)arg;
    const mapn_oracle_sym_shape *sh = J->sh;
    float *scratch = (float *)malloc(sizeof(float) * (3u * SYM_IB + (size_t)sh->waves * 3u * SYM_IB + 2u * sh->waves * 192u));
    if (!scratch) return NULL;
    const uint32_t items = (J->shard_nbl ? J->shard_nbl : sh->nb) * sh->parts;
    for (uint32_t it = J->tid; it < items; it += J->nthreads) sym_workgroup(J, J->a0 + it / sh->parts, it % sh->parts, scratch);
    free(scratch);
    return NULL;
}

/* sym_reduce_integrate_kernel */
static void *sym_reduce_worker(void *arg)
{
    const sym_job *J = (const sym_job *)arg;
    const mapn_oracle_sym_shape *sh = J->sh;
    const uint32_t nb = sh->nb, D = (nb - 1u) / 2u, half = (nb & 1u) ? 0u : nb / 2u;
    const uint32_t g0 = J->win[0], g1 = J->win[1], gs0 = g0 ? g0 : 1u;
    const uint32_t *splits = J->tab + sh->sets * (sh->parts * sh->waves + 1u);
    const size_t np = (size_t)nb * SYM_IB;
    for (uint32_t i = J->tid; i < J->n; i += J->nthreads) {
        const uint32_t a = i / SYM_IB, jb = i >> 6, l = i & 63u, tt = jb % SYM_JPI;
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        if (!J->first_window) { ax = J->acc[i]; ay = J->acc[np + i]; az = J->acc[2 * np + i]; }
        for (uint32_t s = 0; s < sh->parts; s++) {
            const float *row = J->arow + ((size_t)a * sh->parts + s) * 3u * SYM_IB + (i - a * SYM_IB);
            ax = ax + row[0]; ay = ay + row[SYM_IB]; az = az + row[2u * SYM_IB];
        }
        const uint32_t gend = (g1 == D + 2u && sym_runs_half(a, half)) ? D + 1u : g1;
        for (uint32_t g = gs0; g < gend; g += 8u) {
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t gu = g + u;
                float vx = 0.0f, vy = 0.0f, vz = 0.0f, hx = 0.0f, hy = 0.0f, hz = 0.0f;
                if (gu < gend) {
                    const uint32_t d = gu <= D ? gu : half, ap = a >= d ? a - d : a + nb - d;
                    const float *r0 = J->brow + ((size_t)jb * sh->brows + (gu - gs0)) * 192u;
                    vx = r0[l]; vy = r0[64 + l]; vz = r0[128 + l];
                    const uint32_t sp = (splits + (size_t)((sym_runs_half(ap, half) ? 0u : 1u) + (sh->sets > 2u ? 2u * (ap & 7u) : 0u)) * sh->max_meetings)[(gu - g0) * SYM_JPI + tt];
                    if (sp != SYM_NONE) {
                        const float *h0 = J->brow1 + ((size_t)ap * sh->parts + sp) * 192u;
                        hx = h0[l]; hy = h0[64 + l]; hz = h0[128 + l];
                    }
                }
                ax = ax + vx; ay = ay + vy; az = az + vz;
                ax = ax + hx; ay = ay + hy; az = az + hz;
            }
        }
        if (!J->last_window) { J->acc[i] = ax; J->acc[np + i] = ay; J->acc[2 * np + i] = az; continue; }
        integrate_fused(J->old_pos + 4 * (size_t)i, J->old_vel + 3 * (size_t)i, ax * J->p->mass, ay * J->p->mass, az * J->p->mass,
                        J->p, J->new_pos + 4 * (size_t)i, J->new_vel + 3 * (size_t)i);
    }
    return NULL;
}

static int sym_run(void *(*fn)(void *), sym_job *proto, int threads)
{
    sym_job *jobs = (sym_job *)calloc((size_t)threads, sizeof(sym_job));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    for (int t = 0; t < threads; t++) {
        jobs[t] = *proto;
        jobs[t].tid = (uint32_t)t; jobs[t].nthreads = (uint32_t)threads;
        if (t > 0 && pthread_create(&th[t], NULL, fn, &jobs[t]) != 0) { fn(&jobs[t]); th[t] = 0; }
    }
    fn(&jobs[0]);
    for (int t = 1; t < threads; t++) if (th[t]) pthread_join(th[t], NULL);
    free(jobs); free(th);
    return 0;
}

/* One whole-N all-pairs step in the symmetric kernel's order.  windows: [shape->windows][4], tables:
 * [shape->windows][shape->table_stride] exactly as mapn_get_sym_plan / mapn_sym_plan_describe return them. */
int mapn_oracle_step_all_pairs_sym(const float *old_pos, const float *old_vel, float *new_pos, float *new_vel, uint32_t n,
                                   const mapn_oracle_params *p, int threads, const mapn_oracle_sym_shape *shape,
                                   const uint32_t *windows, const uint32_t *tables)
{
    if (!shape || !windows || !tables || shape->nb != (n + SYM_IB - 1u) / SYM_IB || shape->waves == 0 || shape->parts == 0 || (shape->sets != 2u && shape->sets != 16u)) return -2;
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    const size_t np = (size_t)shape->nb * SYM_IB;
    float *arow = (float *)malloc(sizeof(float) * 3u * SYM_IB * shape->nb * shape->parts);
    float *brow = (float *)malloc(sizeof(float) * 192u * (np / 64u) * shape->brows);
    float *brow1 = (float *)malloc(sizeof(float) * 192u * shape->nb * shape->parts);
    float *acc = (float *)malloc(sizeof(float) * 3u * np);
    int rc = (arow && brow && brow1 && acc) ? 0 : -1;
    for (uint32_t k = 0; k < shape->windows && rc == 0; k++) {
        sym_job J = {old_pos, old_vel, new_pos, new_vel, n, p, shape, windows + 4u * k, tables + (size_t)k * shape->table_stride,
                     arow, brow, brow1, acc, k == 0, k + 1u == shape->windows, 0, 0, 0, 0};
        const uint32_t items = shape->nb * shape->parts;
        rc = sym_run(sym_force_worker, &J, (uint32_t)threads > items ? (int)items : threads);
        if (rc == 0) rc = sym_run(sym_reduce_worker, &J, threads);
    }
    free(arow); free(brow); free(brow1); free(acc);
    return rc;
}

/* =================================================================================================
 * ORDER_MATCHED_SPLIT: the device's PARTIALLY ACTIVE step in its split form (csrc/mapn_sym_host.cpp, enqueue_sym_split;
 * Compute.cpp:1041: bodies [0, active) advance, the frozen ones [active, N) still exert force).  Restated, in the device's order:
 *   * per active body first what the FROZEN bodies do to it: the one-sided kernel's partial rows over the j-range [active, N) --
 *     its ceil((N - active) / 64) tiles cut into S = waves * sb chunks (chunk_tiles), a chunk one fma chain from zero over ascending
 *     j, a row = its `waves` chunk sums added in ascending order to zero -- the rows added in ascending order to zero
 *     (sym_reduce_integrate_kernel, p.extra);
 *   * then the symmetric plan of a job of `active` bodies (the bodies past it are the far-away stand-ins), window by window, exactly
 *     as ORDER_MATCHED_SYM, the frozen sum standing where the running sum of "earlier windows" stands;
 *   * total * mass, fused integrator; bodies [active, N) of the new buffers are left as they are.
 * ================================================================================================= */
typedef struct {
    const float *old_pos;
    uint32_t n_total, n_active, waves, sb;
    float soft2;
    float *acc;                 /* [3][np] */
    size_t np;
    uint32_t tid, nthreads;
    uint32_t j_first, j_count;  /* the frozen j-range (unsharded: [n_active, n_total); sharded: the frozen bodies ONE rank owns) */
} frozen_job;

__attribute__((target_clones("avx512f", "fma", "default")))
static void frozen_block(const frozen_job *F, uint32_t b0, uint32_t nbod)
{
    float xi[IB], yi[IB], zi[IB], tx[IB], ty[IB], tz[IB], wx[IB], wy[IB], wz[IB], cx[IB], cy[IB], cz[IB];
    const float soft2 = F->soft2;
    for (uint32_t k = 0; k < IB; k++) {
        uint32_t i = b0 + (k < nbod ? k : 0);
        xi[k] = F->old_pos[4 * (size_t)i + 0]; yi[k] = F->old_pos[4 * (size_t)i + 1]; zi[k] = F->old_pos[4 * (size_t)i + 2];
        tx[k] = ty[k] = tz[k] = 0.0f;
    }
    const uint32_t waves = F->waves, sb = F->sb, S = waves * sb;
    const uint32_t j_first = F->j_first, j_count = F->j_count;
    const uint32_t tiles = (j_count + 63u) / 64u, base = tiles / S, rem = tiles % S;
    for (uint32_t row = 0; row < sb; row++) {
        for (int k = 0; k < IB; k++) wx[k] = wy[k] = wz[k] = 0.0f;
        for (uint32_t w = 0; w < waves; w++) {
            const uint32_t c = row * waves + w;
            const uint32_t t0 = c * base + (c < rem ? c : rem), t1 = t0 + base + (c < rem ? 1u : 0u);
            uint32_t j0 = t0 * 64u, j1 = t1 * 64u;
            if (j1 > j_count) j1 = j_count;
            for (int k = 0; k < IB; k++) cx[k] = cy[k] = cz[k] = 0.0f;
            const float *pj = F->old_pos + 4 * ((size_t)j_first + j0);
            for (uint32_t j = j0; j < j1; j++, pj += 4) {
                const float xj = pj[0], yj = pj[1], zj = pj[2];
#pragma GCC ivdep
                for (int k = 0; k < IB; k++) {
                    float dx = xj - xi[k], dy = yj - yi[k], dz = zj - zi[k];
                    float d = __builtin_fmaf(dx, dx, soft2);
                    d = __builtin_fmaf(dy, dy, d);
                    d = __builtin_fmaf(dz, dz, d);
                    float inv = 1.0f / sqrtf(d);
                    float inv3 = inv * inv * inv;
                    cx[k] = __builtin_fmaf(dx, inv3, cx[k]);
                    cy[k] = __builtin_fmaf(dy, inv3, cy[k]);
                    cz[k] = __builtin_fmaf(dz, inv3, cz[k]);
                }
            }
            for (int k = 0; k < IB; k++) { wx[k] = wx[k] + cx[k]; wy[k] = wy[k] + cy[k]; wz[k] = wz[k] + cz[k]; }
        }
        for (int k = 0; k < IB; k++) { tx[k] = tx[k] + wx[k]; ty[k] = ty[k] + wy[k]; tz[k] = tz[k] + wz[k]; }
    }
    for (uint32_t k = 0; k < nbod; k++) {
        F->acc[b0 + k] = tx[k]; F->acc[F->np + b0 + k] = ty[k]; F->acc[2 * F->np + b0 + k] = tz[k];
    }
}

static void *frozen_worker(void *arg)
{
    const frozen_job *F = (const frozen_job *)arg;
    const uint32_t nblocks = (F->n_active + IB - 1) / IB;
    for (uint32_t b = F->tid; b < nblocks; b += F->nthreads) {
        const uint32_t b0 = b * IB, left = F->n_active - b0;
        frozen_block(F, b0, left < IB ? left : IB);
    }
    return NULL;
}

/* shape / windows / tables: the plan of the ACTIVE bodies (mapn_get_split_plan); frozen_waves, frozen_sb: the one-sided launch's shape */
int mapn_oracle_step_all_pairs_sym_split(const float *old_pos, const float *old_vel, float *new_pos, float *new_vel, uint32_t n_total,
                                         uint32_t n_active, const mapn_oracle_params *p, int threads, const mapn_oracle_sym_shape *shape,
                                         const uint32_t *windows, const uint32_t *tables, uint32_t frozen_waves, uint32_t frozen_sb)
{
    if (!shape || !windows || !tables || n_active == 0 || n_active >= n_total || shape->nb != (n_active + SYM_IB - 1u) / SYM_IB ||
        shape->waves == 0 || shape->parts == 0 || (shape->sets != 2u && shape->sets != 16u) || frozen_waves == 0 || frozen_sb == 0) return -2;
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    const size_t np = (size_t)shape->nb * SYM_IB;
    float *arow = (float *)malloc(sizeof(float) * 3u * SYM_IB * shape->nb * shape->parts);
    float *brow = (float *)malloc(sizeof(float) * 192u * (np / 64u) * shape->brows);
    float *brow1 = (float *)malloc(sizeof(float) * 192u * shape->nb * shape->parts);
    float *acc = (float *)calloc(3u * np, sizeof(float));
    int rc = (arow && brow && brow1 && acc) ? 0 : -1;
    if (rc == 0) {
        /* (1) the frozen bodies' rows, summed: the start value of every active body's sum */
        const uint32_t nblocks = (n_active + IB - 1) / IB;
        const int ft = (uint32_t)threads > nblocks ? (int)nblocks : threads;
        frozen_job *jobs = (frozen_job *)calloc((size_t)ft, sizeof(frozen_job));
        pthread_t *th = (pthread_t *)calloc((size_t)ft, sizeof(pthread_t));
        if (!jobs || !th) rc = -1;
        for (int t = 0; t < ft && rc == 0; t++) {
            jobs[t] = (frozen_job){old_pos, n_total, n_active, frozen_waves, frozen_sb, p->soft2, acc, np, (uint32_t)t, (uint32_t)ft, n_active, n_total - n_active};
            if (t > 0 && pthread_create(&th[t], NULL, frozen_worker, &jobs[t]) != 0) { frozen_worker(&jobs[t]); th[t] = 0; }
        }
        if (rc == 0) {
            frozen_worker(&jobs[0]);
            for (int t = 1; t < ft; t++) if (th[t]) pthread_join(th[t], NULL);
        }
        free(jobs); free(th);
    }
    /* (2) the active bodies among themselves: the symmetric plan of a job of n_active bodies, the frozen sum as the running sum */
    for (uint32_t k = 0; k < shape->windows && rc == 0; k++) {
        sym_job J = {old_pos, old_vel, new_pos, new_vel, n_active, p, shape, windows + 4u * k, tables + (size_t)k * shape->table_stride,
                     arow, brow, brow1, acc, 0, k + 1u == shape->windows, 0, 0, 0, 0};
        const uint32_t items = shape->nb * shape->parts;
        rc = sym_run(sym_force_worker, &J, (uint32_t)threads > items ? (int)items : threads);
        if (rc == 0) rc = sym_run(sym_reduce_worker, &J, threads);
    }
    free(arow); free(brow); free(brow1); free(acc);
    return rc;
}

/* =================================================================================================
 * ORDER_MATCHED_SHARDED: the device's SYMMETRIC step SHARDED over ranks (gather algorithms 4 / 5 / 6; csrc/mapn_sym.hip,
 * sym_shard_exchange_kernel), all ranks restated in one process.  Rank p owns the blocks [a0, a0 + nbl) and
 *   * runs their meetings under ITS plan (force_sym_kernel with a0 / shard_nbl; one window): a-rows per (local block, part), reaction
 *     rows per (J-block, local block), head rows per (local block, part) -- sym_workgroup above with J->a0, J->shard_nbl;
 *   * SENDS, per destination rank q and body t of q: the rows of its blocks that met the body's block, local blocks ascending, a
 *     meeting's row then its head row (zeros where there is none), added to zero;
 *   * INTEGRATES its bodies: G partial sums of the a-rows -- parts [P g / G, P (g + 1) / G) ascending from zero -- added in ascending g
 *     to zero, then the rows received, nearest sender first (this rank, rank - 1, rank - 2, ... mod world; a zero for a rank that sends
 *     nothing, up to 16 places like the device), then total * mass and the fused integrator.
 * What is left against the device is v_rsq_f32 alone; the transport (pushed / pulled positions, RCCL) changes no bit.
 * ================================================================================================= */
#define SHARD_MAX_RANKS 16u

static inline uint32_t sym_group_of(uint32_t a, uint32_t b, uint32_t nb, uint32_t half)
{
    const uint32_t d = b >= a ? b - a : b + nb - a, D = (nb - 1u) / 2u;
    return (d >= 1u && d <= D) ? d : (half && d == half && sym_runs_half(a, half)) ? D + 1u : 0u;
}

/* shapes[r], windows + 4 r, tables + table_offset[r]: rank r's plan (mapn_get_sym_plan on that rank); G: threads per body of the exchange
 * launch (8 up to 16 384 bodies per rank, 4 up to 65 536, else 1: exchange_threads_per_body) */
int mapn_oracle_step_all_pairs_sym_sharded(const float *old_pos, const float *old_vel, float *new_pos, float *new_vel, uint32_t n,
                                           const mapn_oracle_params *p, int threads, uint32_t world, const mapn_oracle_sym_shape *shapes,
                                           const uint32_t *windows, const uint32_t *tables, const uint64_t *table_offset, uint32_t G, int32_t only_rank)
{
    /* only_rank >= 0: that rank ALONE, receiving nothing from the others (the device's loopback hook MAPN_P2P_LOOPBACK=2: its bodies get the
     * forces of its blocks' meetings plus the reactions of meetings between two of its own blocks); the other ranks' bodies are left as they are */
    if (!shapes || !windows || !tables || !table_offset || world < 2u || world > SHARD_MAX_RANKS || n % world || (n / world) % SYM_IB || G == 0u) return -2;
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    const uint32_t count = n / world, nb = n / SYM_IB, nbl = count / SYM_IB, half = (nb & 1u) ? 0u : nb / 2u;
    for (uint32_t r = 0; r < world; r++)
        if (shapes[r].nb != nb || shapes[r].windows != 1u || shapes[r].waves == 0 || shapes[r].parts == 0 || (shapes[r].sets != 2u && shapes[r].sets != 16u)) return -2;
    /* recv[q][r][body of q]: what rank r sends rank q */
    float *recv = (float *)calloc((size_t)world * world * count * 3u, sizeof(float));
    uint32_t sends[SHARD_MAX_RANKS][SHARD_MAX_RANKS];          /* sends[r][q] != 0: rank r produces reactions for rank q */
    memset(sends, 0, sizeof sends);
    for (uint32_t a = 0; a < nb; a++)
        for (uint32_t b = 0; b < nb; b++)
            if (sym_group_of(a, b, nb, half)) sends[a / nbl][b / nbl] = 1u;
    size_t arow_max = 0;
    for (uint32_t r = 0; r < world; r++) if (shapes[r].parts > arow_max) arow_max = shapes[r].parts;
    float **arows = (float **)calloc(world, sizeof(float *));
    float *brow = (float *)malloc(sizeof(float) * 192u * (size_t)(n / 64u) * nbl);
    float *brow1 = (float *)malloc(sizeof(float) * 192u * (size_t)nbl * arow_max);
    int rc = (recv && arows && brow && brow1) ? 0 : -1;
    for (uint32_t r = 0; r < world && rc == 0; r++) {
        if (only_rank >= 0 && r != (uint32_t)only_rank) continue;
        const mapn_oracle_sym_shape *sh = &shapes[r];
        const uint32_t a0 = r * nbl;
        arows[r] = (float *)malloc(sizeof(float) * 3u * SYM_IB * (size_t)nbl * sh->parts);
        if (!arows[r]) { rc = -1; break; }
        sym_job J = {old_pos, old_vel, new_pos, new_vel, n, p, sh, windows + 4u * r, tables + table_offset[r], arows[r], brow, brow1, NULL, 1, 1, 0, 0, a0, nbl};
        const uint32_t items = nbl * sh->parts;
        rc = sym_run(sym_force_worker, &J, (uint32_t)threads > items ? (int)items : threads);
        if (rc) break;
        /* SEND: per destination q and body t of q */
        const uint32_t *splits = J.tab + sh->sets * (sh->parts * sh->waves + 1u);
        for (uint32_t q = 0; q < world; q++) {
            if (!sends[r][q]) continue;
            for (uint32_t jl = 0; jl < count; jl++) {
                const uint32_t t = q * count + jl, b = t / SYM_IB, jb = t >> 6, tt = jb % SYM_JPI, l = t & 63u;
                float fx = 0.0f, fy = 0.0f, fz = 0.0f;
                for (uint32_t la0 = 0; la0 < nbl; la0 += 8u)
                    for (uint32_t u = 0; u < 8u; u++) {                    /* batches of eight like the device: a missing row is a zero that is still added */
                        const uint32_t la = la0 + u;
                        float vx = 0.0f, vy = 0.0f, vz = 0.0f, hx = 0.0f, hy = 0.0f, hz = 0.0f;
                        const uint32_t gg = la < nbl ? sym_group_of(a0 + la, b, nb, half) : 0u;
                        if (gg) {
                            const float *r0 = brow + ((size_t)jb * nbl + la) * 192u;
                            vx = r0[l]; vy = r0[64 + l]; vz = r0[128 + l];
                            const uint32_t set = (sym_runs_half(a0 + la, half) ? 0u : 1u) + (sh->sets > 2u ? 2u * (la & 7u) : 0u);
                            const uint32_t sp = (splits + (size_t)set * sh->max_meetings)[gg * SYM_JPI + tt];
                            if (sp != SYM_NONE) {
                                const float *h0 = brow1 + ((size_t)la * sh->parts + sp) * 192u;
                                hx = h0[l]; hy = h0[64 + l]; hz = h0[128 + l];
                            }
                        }
                        fx = fx + vx; fy = fy + vy; fz = fz + vz;
                        fx = fx + hx; fy = fy + hy; fz = fz + hz;
                    }
                float *dst = recv + (((size_t)q * world + r) * count + jl) * 3u;
                dst[0] = fx; dst[1] = fy; dst[2] = fz;
            }
        }
    }
    /* INTEGRATE: every rank its own bodies */
    for (uint32_t r = 0; r < world && rc == 0; r++) {
        if (only_rank >= 0 && r != (uint32_t)only_rank) continue;
        const mapn_oracle_sym_shape *sh = &shapes[r];
        for (uint32_t il = 0; il < count; il++) {
            const uint32_t la = il / SYM_IB, e = il - la * SYM_IB, i = r * count + il;
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            for (uint32_t g = 0; g < G; g++) {
                const uint32_t s0 = (uint32_t)(((uint64_t)sh->parts * g) / G), s1 = (uint32_t)(((uint64_t)sh->parts * (g + 1u)) / G);
                float px = 0.0f, py = 0.0f, pz = 0.0f;
                for (uint32_t s = s0; s < s1; s++) {
                    const float *row = arows[r] + ((size_t)la * sh->parts + s) * 3u * SYM_IB + e;
                    px = px + row[0]; py = py + row[SYM_IB]; pz = pz + row[2u * SYM_IB];
                }
                ax = ax + px; ay = ay + py; az = az + pz;
            }
            for (uint32_t k = 0; k < SHARD_MAX_RANKS; k++) {
                float rx = 0.0f, ry = 0.0f, rz = 0.0f;
                if (k < world) {
                    const uint32_t q = r >= k ? r - k : r + world - k;
                    if (sends[q][r] && (only_rank < 0 || q == r)) {
                        const float *src = recv + (((size_t)r * world + q) * count + il) * 3u;
                        rx = src[0]; ry = src[1]; rz = src[2];
                    }
                }
                ax = ax + rx; ay = ay + ry; az = az + rz;
            }
            integrate_fused(old_pos + 4 * (size_t)i, old_vel + 3 * (size_t)i, ax * p->mass, ay * p->mass, az * p->mass, p,
                            new_pos + 4 * (size_t)i, new_vel + 3 * (size_t)i);
        }
    }
    for (uint32_t r = 0; r < world; r++) free(arows ? arows[r] : NULL);
    free(arows); free(recv); free(brow); free(brow1);
    return rc;
}

/* =================================================================================================
 * ORDER_MATCHED_SHARDED_SPLIT: the device's PARTIALLY ACTIVE step of a SHARDED job (csrc/mapn_sym_host.cpp, enqueue_sym_shard_split;
 * Compute.cpp:1041: the bodies [0, n_active) of the whole job advance, the frozen ones still exert force), all ranks restated in one
 * process.  The active bodies form a ring of nba = ceil(n_active / 1024) blocks; rank r owns count = n / world bodies:
 *   * its ACTIVE ones (its first ac_r) lie in nbl_r = ceil(ac_r / 1024) blocks of the ring from block r * count / 1024: it runs their
 *     meetings under ITS plan (shapes[r]; a job of n_active bodies, the rest far-away stand-ins) exactly as ORDER_MATCHED_SHARDED;
 *   * its FROZEN ones [fz_first_r, (r + 1) count): per active body t of the whole job the one-sided kernel's partial rows over that
 *     j-range (frozen_waves[r] x frozen_sb[r] chunks: ORDER_MATCHED_SPLIT's frozen rows with this j-range), added ascending to zero;
 *   * SENDS per destination body: the frozen sum (zero if it owns no frozen body), then the rows of its blocks that met the body's block
 *     as ORDER_MATCHED_SHARDED adds them;
 *   * INTEGRATES its ac_r active bodies: the G partial sums of its a-rows, then the rows received nearest sender first; the frozen
 *     bodies of the new buffers are left as they are.
 * shapes[r].nb == 0: rank r owns no active body (no plan).  frozen_waves[r] == 0: it owns no frozen body.
 * ================================================================================================= */
int mapn_oracle_step_all_pairs_sym_sharded_split(const float *old_pos, const float *old_vel, float *new_pos, float *new_vel, uint32_t n, uint32_t n_active,
                                                 const mapn_oracle_params *p, int threads, uint32_t world, const mapn_oracle_sym_shape *shapes,
                                                 const uint32_t *windows, const uint32_t *tables, const uint64_t *table_offset, uint32_t G,
                                                 const uint32_t *frozen_waves, const uint32_t *frozen_sb)
{
    if (!shapes || !windows || !tables || !table_offset || !frozen_waves || !frozen_sb || world < 2u || world > SHARD_MAX_RANKS || n % world ||
        (n / world) % SYM_IB || G == 0u || n_active == 0u || n_active >= n || n_active % 64u) return -2;
    if (threads <= 0) threads = mapn_oracle_hardware_threads();
    const uint32_t count = n / world, cblk = count / SYM_IB, nba = (n_active + SYM_IB - 1u) / SYM_IB, half = (nba & 1u) ? 0u : nba / 2u;
    uint32_t ac[SHARD_MAX_RANKS], nbl[SHARD_MAX_RANKS], fz_first[SHARD_MAX_RANKS], fz_count[SHARD_MAX_RANKS], nbl_max = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint32_t first = r * count, hi = first + count < n_active ? first + count : n_active;
        ac[r] = hi > first ? hi - first : 0u;
        nbl[r] = (ac[r] + SYM_IB - 1u) / SYM_IB;
        fz_first[r] = first + ac[r];
        fz_count[r] = count - ac[r];
        if (nbl[r] > nbl_max) nbl_max = nbl[r];
        if (nbl[r] && (shapes[r].nb != nba || shapes[r].windows != 1u || shapes[r].waves == 0 || shapes[r].parts == 0 || (shapes[r].sets != 2u && shapes[r].sets != 16u))) return -2;
        if (fz_count[r] && (frozen_waves[r] == 0u || frozen_sb[r] == 0u)) return -2;
    }
    const size_t np = (size_t)nba * SYM_IB;
    /* recv[q][r][body of q]: what rank r sends rank q */
    float *recv = (float *)calloc((size_t)world * world * count * 3u, sizeof(float));
    uint32_t sends[SHARD_MAX_RANKS][SHARD_MAX_RANKS];
    memset(sends, 0, sizeof sends);
    for (uint32_t a = 0; a < nba; a++)
        for (uint32_t b = 0; b < nba; b++)
            if (sym_group_of(a, b, nba, half)) sends[a / cblk][b / cblk] = 1u;
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t q = 0; q < world; q++)
            if (fz_count[r] && ac[q]) sends[r][q] = 1u;
    size_t parts_max = 1;
    for (uint32_t r = 0; r < world; r++) if (nbl[r] && shapes[r].parts > parts_max) parts_max = shapes[r].parts;
    float **arows = (float **)calloc(world, sizeof(float *));
    float *brow = (float *)malloc(sizeof(float) * 192u * (np / 64u) * (nbl_max ? nbl_max : 1u));
    float *brow1 = (float *)malloc(sizeof(float) * 192u * (size_t)(nbl_max ? nbl_max : 1u) * parts_max);
    float *facc = (float *)malloc(sizeof(float) * 3u * np);
    int rc = (recv && arows && brow && brow1 && facc) ? 0 : -1;
    for (uint32_t r = 0; r < world && rc == 0; r++) {
        const mapn_oracle_sym_shape *sh = &shapes[r];
        const uint32_t a0 = r * cblk;
        sym_job J = {old_pos, old_vel, new_pos, new_vel, n_active, p, sh, windows + 4u * r, tables + table_offset[r], NULL, brow, brow1, NULL, 1, 1, 0, 0, a0, nbl[r]};
        if (nbl[r]) {
            arows[r] = (float *)malloc(sizeof(float) * 3u * SYM_IB * (size_t)nbl[r] * sh->parts);
            if (!arows[r]) { rc = -1; break; }
            J.arow = arows[r];
            const uint32_t items = nbl[r] * sh->parts;
            rc = sym_run(sym_force_worker, &J, (uint32_t)threads > items ? (int)items : threads);
            if (rc) break;
        }
        if (fz_count[r]) {
            /* what rank r's frozen bodies do to every active body of the job */
            const uint32_t nblocks = (n_active + IB - 1) / IB;
            const int ft = (uint32_t)threads > nblocks ? (int)nblocks : threads;
            frozen_job *jobs = (frozen_job *)calloc((size_t)ft, sizeof(frozen_job));
            pthread_t *th = (pthread_t *)calloc((size_t)ft, sizeof(pthread_t));
            if (!jobs || !th) { free(jobs); free(th); rc = -1; break; }
            for (int t = 0; t < ft; t++) {
                jobs[t] = (frozen_job){old_pos, n, n_active, frozen_waves[r], frozen_sb[r], p->soft2, facc, np, (uint32_t)t, (uint32_t)ft, fz_first[r], fz_count[r]};
                if (t > 0 && pthread_create(&th[t], NULL, frozen_worker, &jobs[t]) != 0) { frozen_worker(&jobs[t]); th[t] = 0; }
            }
            frozen_worker(&jobs[0]);
            for (int t = 1; t < ft; t++) if (th[t]) pthread_join(th[t], NULL);
            free(jobs); free(th);
        }
        /* SEND: per destination q and ACTIVE body t of q */
        const uint32_t *splits = nbl[r] ? J.tab + sh->sets * (sh->parts * sh->waves + 1u) : NULL;
        for (uint32_t q = 0; q < world; q++) {
            if (!sends[r][q]) continue;
            for (uint32_t jl = 0; jl < ac[q]; jl++) {
                const uint32_t t = q * count + jl, b = t / SYM_IB, jb = t >> 6, tt = jb % SYM_JPI, l = t & 63u;
                float fx = 0.0f, fy = 0.0f, fz = 0.0f;
                if (fz_count[r]) { fx = facc[t]; fy = facc[np + t]; fz = facc[2 * np + t]; }
                for (uint32_t la0 = 0; la0 < nbl[r]; la0 += 8u)
                    for (uint32_t u = 0; u < 8u; u++) {                    /* batches of eight like the device: a missing row is a zero that is still added */
                        const uint32_t la = la0 + u;
                        float vx = 0.0f, vy = 0.0f, vz = 0.0f, hx = 0.0f, hy = 0.0f, hz = 0.0f;
                        const uint32_t gg = la < nbl[r] ? sym_group_of(a0 + la, b, nba, half) : 0u;
                        if (gg) {
                            const float *r0 = brow + ((size_t)jb * nbl[r] + la) * 192u;
                            vx = r0[l]; vy = r0[64 + l]; vz = r0[128 + l];
                            const uint32_t set = (sym_runs_half(a0 + la, half) ? 0u : 1u) + (sh->sets > 2u ? 2u * (la & 7u) : 0u);
                            const uint32_t sp = (splits + (size_t)set * sh->max_meetings)[gg * SYM_JPI + tt];
                            if (sp != SYM_NONE) {
                                const float *h0 = brow1 + ((size_t)la * sh->parts + sp) * 192u;
                                hx = h0[l]; hy = h0[64 + l]; hz = h0[128 + l];
                            }
                        }
                        fx = fx + vx; fy = fy + vy; fz = fz + vz;
                        fx = fx + hx; fy = fy + hy; fz = fz + hz;
                    }
                float *dst = recv + (((size_t)q * world + r) * count + jl) * 3u;
                dst[0] = fx; dst[1] = fy; dst[2] = fz;
            }
        }
    }
    /* INTEGRATE: every rank its own ACTIVE bodies */
    for (uint32_t r = 0; r < world && rc == 0; r++) {
        const mapn_oracle_sym_shape *sh = &shapes[r];
        for (uint32_t il = 0; il < ac[r]; il++) {
            const uint32_t la = il / SYM_IB, e = il - la * SYM_IB, i = r * count + il;
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            for (uint32_t g = 0; g < G; g++) {
                const uint32_t s0 = (uint32_t)(((uint64_t)sh->parts * g) / G), s1 = (uint32_t)(((uint64_t)sh->parts * (g + 1u)) / G);
                float px = 0.0f, py = 0.0f, pz = 0.0f;
                for (uint32_t s = s0; s < s1; s++) {
                    const float *row = arows[r] + ((size_t)la * sh->parts + s) * 3u * SYM_IB + e;
                    px = px + row[0]; py = py + row[SYM_IB]; pz = pz + row[2u * SYM_IB];
                }
                ax = ax + px; ay = ay + py; az = az + pz;
            }
            for (uint32_t k = 0; k < SHARD_MAX_RANKS; k++) {
                float rx = 0.0f, ry = 0.0f, rz = 0.0f;
                if (k < world) {
                    const uint32_t q = r >= k ? r - k : r + world - k;
                    if (sends[q][r]) {
                        const float *src = recv + (((size_t)r * world + q) * count + il) * 3u;
                        rx = src[0]; ry = src[1]; rz = src[2];
                    }
                }
                ax = ax + rx; ay = ay + ry; az = az + rz;
            }
            integrate_fused(old_pos + 4 * (size_t)i, old_vel + 3 * (size_t)i, ax * p->mass, ay * p->mass, az * p->mass, p,
                            new_pos + 4 * (size_t)i, new_vel + 3 * (size_t)i);
        }
    }
    for (uint32_t r = 0; r < world; r++) free(arows ? arows[r] : NULL);
    free(arows); free(recv); free(brow); free(brow1); free(facc);
    return rc;
}
