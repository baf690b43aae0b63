/* Oracle helper (TEST INFRASTRUCTURE ONLY): the per-sample loops of the reference's HASPI that are
 * numba-jitted there (pyHASPI/pyhaspi2.py:843-861 eb_CosSinCF, :1028-1078 eb_IHCadapt) and the
 * sample loop of resampy's resample_f (the kaiser_best resampler behind librosa.resample,
 * pyhaspi2.py:815), restated in plain C so that the CPU oracle and the CPU baseline run at
 * compiled speed as the reference does with numba.  Built by oracle/csrc/Makefile with gcc. */
#include <math.h>

/* pyhaspi2.py:843-861 */
void eb_cos_sin_cf(long npts, double fs, double cf, double* coscf, double* sincf) {
    const double tpt = 2.0 * M_PI / fs;
    const double cn = cos(tpt * cf), sn = sin(tpt * cf);
    double cold = 1.0, sold = 0.0;
    coscf[0] = cold;
    sincf[0] = sold;
    for (long n = 1; n < npts; ++n) {
        const double arg = cold * cn + sold * sn;
        sold = sold * cn - cold * sn;
        cold = arg;
        coscf[n] = cold;
        sincf[n] = sold;
    }
}

/* pyhaspi2.py:1028-1078 (envelope output only; the BM gain is not used by haspi_v2's score) */
void eb_ihc_adapt(const double* xdB, long nsamp, double delta, double fsamp, double* ydB) {
    const double dsmall = 1.0001;
    if (delta < dsmall) delta = dsmall;
    double tau1 = 2, tau2 = 60;
    tau1 = 0.001 * tau1;
    tau2 = 0.001 * tau2;
    const double T = 1 / fsamp;
    const double R1 = 1 / delta;
    const double R2 = 0.5 * (1 - R1);
    const double R3 = R2;
    const double C1 = tau1 * (R1 + R2) / (R1 * R2);
    const double C2 = tau2 / ((R1 + R2) * R3);
    const double a11 = R1 + R2 + R1 * R2 * (C1 / T);
    const double a12 = -R1;
    const double a21 = -R3;
    const double a22 = R2 + R3 + R2 * R3 * (C2 / T);
    const double denom = 1.0 / (a11 * a22 - a21 * a12);
    const double R1inv = 1.0 / R1;
    const double R12C1 = R1 * R2 * (C1 / T);
    const double R23C2 = R2 * R3 * (C2 / T);
    double V1 = 0.0, V2 = 0.0;
    for (long n = 0; n < nsamp; ++n) {
        const double V0 = xdB[n];
        const double b1 = V0 * R2 + R12C1 * V1;
        const double b2 = R23C2 * V2;
        V1 = denom * (a22 * b1 - a12 * b2);
        V2 = denom * (-a21 * b1 + a11 * b2);
        double out = (V0 - V1) * R1inv;
        if (out < 0.0) out = 0.0;
        ydB[n] = out;
    }
}

/* resampy.interpn.resample_f, one channel, float32 in / float32 accumulate as numba does when the
 * input is float32 (y[t] += weight * x[...] rounds to float32 at every step). */
void resample_f32(const float* x, long n_orig, float* y, long n_out, double sample_ratio, const double* interp_win,
                  const double* interp_delta, long nwin, long num_table) {
    const double scale = sample_ratio < 1.0 ? sample_ratio : 1.0;
    const double time_increment = 1.0 / sample_ratio;
    const long index_step = (long)(scale * num_table);
    double time_register = 0.0;
    for (long t = 0; t < n_out; ++t) {
        const long n = (long)time_register;
        double frac = scale * (time_register - n);
        double index_frac = frac * num_table;
        long offset = (long)index_frac;
        double eta = index_frac - offset;
        long i_max = (nwin - offset) / index_step;
        if (n + 1 < i_max) i_max = n + 1;
        float acc = 0.0f;
        for (long i = 0; i < i_max; ++i) {
            const double weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step];
            acc = (float)((double)acc + weight * (double)x[n - i]);
        }
        frac = scale - frac;
        index_frac = frac * num_table;
        offset = (long)index_frac;
        eta = index_frac - offset;
        long k_max = (nwin - offset) / index_step;
        if (n_orig - n - 1 < k_max) k_max = n_orig - n - 1;
        for (long k = 0; k < k_max; ++k) {
            const double weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step];
            acc = (float)((double)acc + weight * (double)x[n + k + 1]);
        }
        y[t] = acc;
        time_register += time_increment;
    }
}
