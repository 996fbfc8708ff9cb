/* ORACLE (test infrastructure; never linked into or called by the product path). */
#ifndef ORACLE_H
#define ORACLE_H

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void orc_fft(double* re, double* im, int n, int sign);
void orc_rfft(const double* x, int n, double* Xr, double* Xi, double* work);
void orc_irfft(const double* Xr, const double* Xi, int n, double* x, double* work);

/* shared WORLD helpers (world.c) */
int orc_mround(double x);
void orc_interp1(const double* x, const double* y, int n, const double* xi, int m, double* yi);
void orc_nuttall(int n, double* w);

#endif
