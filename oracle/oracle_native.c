/* TEST INFRASTRUCTURE ONLY -- C restatement of the two native helpers on GPy's exact-GP hot path.
 *
 * oracle_lengthscale_grads: grad[q] = sum_{n,m} tmp[n,m] * (X[n,q]-X2[m,q])^2, the serial triple
 *   loop the reference actually calls for ARD kernels (GPy/kern/src/stationary_cython.pyx:53-62,
 *   reached from GPy/kern/src/stationary.py:237-243).  Kept single-threaded like the reference.
 * oracle_symmetrify: copy one triangle of a row-major n x n matrix onto the other
 *   (GPy/util/linalg_cython.pyx:9-18, reached from GPy/util/linalg.py:356-370).
 *
 * Built by oracle/Makefile into oracle/_build/liboracle_native.so; used only by oracle/gp_oracle.py
 * (tests, smoke(), bench.py's cpu_baseline leg).  Never linked into the product library.
 */
void oracle_lengthscale_grads(long N, long M, long Q, const double *tmp, const double *X,
                              const double *X2, double *grad)
{
    for (long q = 0; q < Q; ++q) {
        double g = 0.0;
        for (long n = 0; n < N; ++n) {
            const double xn = X[n * Q + q];
            const double *t = tmp + n * M;
            for (long m = 0; m < M; ++m) {
                const double d = xn - X2[m * Q + q];
                g += t[m] * d * d;
            }
        }
        grad[q] += g;
    }
}

void oracle_symmetrify(long n, double *A, int upper)
{
    if (upper) {            /* upper triangle is the source */
        for (long i = 0; i < n; ++i)
            for (long j = 0; j < i; ++j)
                A[i * n + j] = A[j * n + i];
    } else {                /* lower triangle is the source */
        for (long i = 0; i < n; ++i)
            for (long j = 0; j < i; ++j)
                A[j * n + i] = A[i * n + j];
    }
}
