/* Round-6 probe (VERDICT r5 task 1): how far do delays move when a level's TOP sample, which sits on model node z[kz+1] within the
 * residual delta of the 3-iteration crossing (losreader.py:724-731,817-819), is evaluated AT the node instead of at its true height?
 * Includes the oracle's C restatement (test infrastructure) and adds a variant of its march that records, per ray:
 *   out[0] = sum over node tops of weight * |dN/dz|-ish exact difference (the node-mode delay minus the faithful one), hydro
 *   out[1] = the same, wet;   out[2] = max |delta| (m) over the ray's node tops;   out[3] = delta at the last node top
 * Build: gcc -O2 -fopenmp -shared -fPIC -ffp-contract=off tools/probes/nodetop_probe.c -o tools/probes/nodetop_probe.so -lm */
#include "../../oracle/oracle_c.c"

void probe_nodetop(const double* lat, const double* lon, const double* los, int64_t n, double ht,
                   const double* lo, const double* hi, int K, const int* nparts,
                   const double* ys, int ny, const double* xs, int nx, const double* zs, int nz, const void* wet, const void* hyd, int dtype,
                   double* out /* [n][4] */) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double o[3], low[3], high[3], cosf = 1.0, dw = 0.0, dh = 0.0, dmax = 0.0, dlast = 0.0;
        lla2ecef(lat[i], lon[i], ht, &o[0], &o[1], &o[2]);
        const double* l = los + 3 * i;
        for (int k = 0; k < K; ++k) {
            if (k == 0) toa(o, l, lo[0], 10, 1.0, low); else { low[0] = high[0]; low[1] = high[1]; low[2] = high[2]; }
            toa(o, l, hi[k], k == 0 ? 10 : 3, cosf, high);
            const double dx = high[0] - low[0], dy = high[1] - low[1], dz = high[2] - low[2];
            const double L = sqrt(dx * dx + dy * dy + dz * dz);
            if (k == 0) cosf = (hi[0] - lo[0]) / L;
            /* the top sample of level k: is hi[k] a model node? */
            int node = 0;
            for (int z = 0; z < nz; ++z) if (zs[z] == hi[k]) node = 1;
            double plon, plat, ph;
            ecef2lla(high[0], high[1], high[2], &plon, &plat, &ph);
            if (node) {
                double vw, vh, nw, nh;
                rgi2(ys, ny, xs, nx, zs, nz, wet, hyd, dtype, plat, plon, ph, &vw, &vh);
                rgi2(ys, ny, xs, nx, zs, nz, wet, hyd, dtype, plat, plon, hi[k], &nw, &nh);
                /* weight: half of this segment's L*1e-6/(np-1); the other half (next segment) is added below with the next L */
                const double wt = 0.5 * (L * 1.0e-6 / (nparts[k] - 1.0));
                dw += wt * (nw - vw); dh += wt * (nh - vh);
                /* next segment's bottom is the same point */
                if (k + 1 < K) {
                    double h2[3];
                    toa(o, l, hi[k + 1], 3, cosf, h2);
                    const double ex = h2[0] - high[0], ey = h2[1] - high[1], ez = h2[2] - high[2];
                    const double L2 = sqrt(ex * ex + ey * ey + ez * ez);
                    const double wt2 = 0.5 * (L2 * 1.0e-6 / (nparts[k + 1] - 1.0));
                    dw += wt2 * (nw - vw); dh += wt2 * (nh - vh);
                }
                const double d = ph - hi[k];
                if (fabs(d) > dmax) dmax = fabs(d);
                dlast = d;
            }
        }
        out[4 * i] = dh; out[4 * i + 1] = dw; out[4 * i + 2] = dmax; out[4 * i + 3] = dlast;
    }
}
