// Host driver around the per-link device math of the SU(3) training kernels.
// stdin: op name, then the operands as doubles; stdout: the result matrices.
//   projsu_vjp : M (9 complex), gy (8)          -> g_M (9 complex)
//   frechet    : B (9 complex), G (9 complex)   -> exp(B), L_exp(B)[G]   (Cayley-Hamilton form, ships)
//   frechet_series : the same pair from the matrix-valued Taylor recursion (kept for A/B)
//   expm       : A (9 complex)                  -> m3_expm(A)
#include <cstdio>
#include <cstring>
#include "su3_train_math.hpp"
using namespace l2q;

static bool read_m3(M3& m) {
  for (int i = 0; i < 9; ++i)
    if (scanf("%lf %lf", &m.re[i], &m.im[i]) != 2) return false;
  return true;
}
static void print_m3(const M3& m) {
  for (int i = 0; i < 9; ++i) printf("%.17g %.17g\n", m.re[i], m.im[i]);
}

int main() {
  char op[32];
  while (scanf("%31s", op) == 1) {
    if (!strcmp(op, "projsu_vjp")) {
      M3 m, g;
      double gy[8];
      if (!read_m3(m)) return 1;
      for (int i = 0; i < 8; ++i)
        if (scanf("%lf", &gy[i]) != 1) return 1;
      m3_projsu_vec8_vjp(g, m, gy);
      print_m3(g);
    } else if (!strcmp(op, "frechet")) {
      M3 b, g, e, l;
      if (!read_m3(b) || !read_m3(g)) return 1;
      m3_expm_frechet(e, l, b, g);
      print_m3(e);
      print_m3(l);
    } else if (!strcmp(op, "frechet_series")) {
      M3 b, g, e, l;
      if (!read_m3(b) || !read_m3(g)) return 1;
      m3_expm_frechet_series(e, l, b, g);
      print_m3(e);
      print_m3(l);
    } else if (!strcmp(op, "expm")) {
      M3 a, e;
      if (!read_m3(a)) return 1;
      m3_expm(e, a);
      print_m3(e);
    } else {
      return 2;
    }
  }
  return 0;
}
