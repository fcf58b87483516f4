/* bls381_oracle.c -- plain-C CPU restatement of the reference's native pairing path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product library never does.
 *
 * PARITY STATUS: field/tower layer pinned by the reference's known answers (tests/golden/reference_kats.json);
 * the Miller loops are PARITY UNPINNED by the reference (its only hot-path test prints without asserting,
 * src/miller_loop_native_optimized.rs:138-145; arkworks sources absent; no Rust toolchain).  This file and
 * oracle/pyref.py are two independent restatements (different number representations: 6x64-bit Montgomery
 * words here, Python big integers there) that must agree bit for bit (tests/test_oracle_c.py).
 *
 * Arithmetic follows what the reference executes through arkworks [third party, ark-ff 0.4.2]: Fp = 6 x u64
 * little-endian limbs in Montgomery form, R = 2^384, every result fully reduced.  Each function cites the
 * reference file:line (relative to /root/reference) whose behaviour it restates.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

/* p, src/fields/bls12_381base.rs:108-113 */
static const fp P = {{0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};
static const uint64_t PINV = 0x89f3fffcfffcfffdull;          /* -p^-1 mod 2^64 */
static const fp R1 = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                       0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};   /* 2^384 mod p */
static const fp R2 = {{0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                       0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull}};   /* 2^768 mod p */
#define BLS_X 0xd201000000010000ull                             /* src/utils/constants.rs:1 */

/* ------------------------------------------------------------------------------------------------ Fp
 * semantics: src/fields/bls12_381base.rs:174-268 (results canonical) */
static int fp_is_zero(const fp* a) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a->l[i]; return o == 0; }
static int fp_eq(const fp* a, const fp* b) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a->l[i] ^ b->l[i]; return o == 0; }

static int geq_p(const uint64_t* t) {
  for (int i = 5; i >= 0; --i) { if (t[i] > P.l[i]) return 1; if (t[i] < P.l[i]) return 0; }
  return 1;
}
static void sub_p(uint64_t* t) {
  u128 b = 0;
  for (int i = 0; i < 6; ++i) { u128 d = (u128)t[i] - P.l[i] - b; t[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static void fp_add(fp* r, const fp* a, const fp* b) {
  u128 c = 0; uint64_t t[6];
  for (int i = 0; i < 6; ++i) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
  if (c || geq_p(t)) sub_p(t);
  memcpy(r->l, t, 48);
}
static void fp_sub(fp* r, const fp* a, const fp* b) {
  u128 bw = 0; uint64_t t[6];
  for (int i = 0; i < 6; ++i) { u128 d = (u128)a->l[i] - b->l[i] - bw; t[i] = (uint64_t)d; bw = (d >> 64) & 1; }
  if (bw) { u128 c = 0; for (int i = 0; i < 6; ++i) { c += (u128)t[i] + P.l[i]; t[i] = (uint64_t)c; c >>= 64; } }
  memcpy(r->l, t, 48);
}
static void fp_neg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); fp_sub(r, &z, a); }
static void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }
/* Montgomery product a*b/2^384 mod p (CIOS) */
static void fp_mul(fp* r, const fp* a, const fp* b) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 6; ++i) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * PINV;
    c = ((u128)m * P.l[0] + t[0]) >> 64;
    for (int j = 1; j < 6; ++j) { c += (u128)m * P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  if (t[6] || geq_p(t)) sub_p(t);
  memcpy(r->l, t, 48);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
/* Fermat inverse (src/fields/bls12_381base.rs:118-125); returns 0 for zero input */
static int fp_inv(fp* r, const fp* a) {
  if (fp_is_zero(a)) { memset(r, 0, sizeof *r); return 0; }
  uint64_t e[6]; memcpy(e, P.l, 48); e[0] -= 2;
  fp acc = R1, base = *a;
  for (int i = 0; i < 384; ++i) {
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
    fp_sqr(&base, &base);
  }
  *r = acc; return 1;
}
static void fp_set_small(fp* r, uint64_t k) { fp t; memset(&t, 0, sizeof t); t.l[0] = k; fp_mul(r, &t, &R2); }

/* ------------------------------------------------------------------------------------------------ Fp2
 * src/fields_as_trees/fq2_target_tree.rs:54-142 */
static void fp2_add(fp2* r, const fp2* a, const fp2* b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2* r, const fp2* a, const fp2* b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2* r, const fp2* a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void fp2_dbl(fp2* r, const fp2* a) { fp2_add(r, a, a); }
static void fp2_mul(fp2* r, const fp2* a, const fp2* b) {          /* :97-115 */
  fp t0, t1, t2, s0, s1;
  fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
  fp_add(&s0, &a->c0, &a->c1); fp_add(&s1, &b->c0, &b->c1); fp_mul(&t2, &s0, &s1);
  fp_sub(&r->c0, &t0, &t1); fp_sub(&t2, &t2, &t0); fp_sub(&r->c1, &t2, &t1);
}
static void fp2_sqr(fp2* r, const fp2* a) {                        /* :80-91 */
  fp s, d, c, t;
  fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_add(&c, &a->c0, &a->c0);
  fp_mul(&t, &c, &a->c1); fp_mul(&r->c0, &s, &d); r->c1 = t;
}
static void fp2_mul_fp(fp2* r, const fp2* a, const fp* k) { fp_mul(&r->c0, &a->c0, k); fp_mul(&r->c1, &a->c1, k); }
static void fp2_conj(fp2* r, const fp2* a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }     /* :117-122 */
static void fp2_mul_xi(fp2* r, const fp2* a) {                     /* :137-142 */
  fp t0, t1; fp_sub(&t0, &a->c0, &a->c1); fp_add(&t1, &a->c0, &a->c1); r->c0 = t0; r->c1 = t1;
}
static int fp2_inv(fp2* r, const fp2* a) {                         /* :66-78 */
  fp n, t, ni; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t);
  if (!fp_inv(&ni, &n)) { memset(r, 0, sizeof *r); return 0; }
  fp_mul(&r->c0, &a->c0, &ni); fp_mul(&t, &a->c1, &ni); fp_neg(&r->c1, &t); return 1;
}
static int fp2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2* a, const fp2* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }

/* ------------------------------------------------------------------------------------------------ Fp6
 * src/fields_as_trees/fq6_target_tree.rs:59-293 */
static void fp6_add(fp6* r, const fp6* a, const fp6* b) { fp2_add(&r->c0, &a->c0, &b->c0); fp2_add(&r->c1, &a->c1, &b->c1); fp2_add(&r->c2, &a->c2, &b->c2); }
static void fp6_sub(fp6* r, const fp6* a, const fp6* b) { fp2_sub(&r->c0, &a->c0, &b->c0); fp2_sub(&r->c1, &a->c1, &b->c1); fp2_sub(&r->c2, &a->c2, &b->c2); }
static void fp6_neg(fp6* r, const fp6* a) { fp2_neg(&r->c0, &a->c0); fp2_neg(&r->c1, &a->c1); fp2_neg(&r->c2, &a->c2); }
static void fp6_mul_v(fp6* r, const fp6* a) {                      /* :219-230 */
  fp2 t; fp2_mul_xi(&t, &a->c2); fp2 a0 = a->c0, a1 = a->c1; r->c0 = t; r->c1 = a0; r->c2 = a1;
}
static void fp6_mul(fp6* r, const fp6* a, const fp6* b) {          /* :172-214 */
  fp2 aa, bb, cc, t1, t2, t3, s, u;
  fp2_mul(&aa, &a->c0, &b->c0); fp2_mul(&bb, &a->c1, &b->c1); fp2_mul(&cc, &a->c2, &b->c2);
  fp2_add(&s, &b->c1, &b->c2); fp2_add(&u, &a->c1, &a->c2); fp2_mul(&t1, &s, &u);
  fp2_sub(&t1, &t1, &bb); fp2_sub(&t1, &t1, &cc); fp2_mul_xi(&t1, &t1); fp2_add(&t1, &t1, &aa);
  fp2_add(&s, &b->c0, &b->c2); fp2_add(&u, &a->c0, &a->c2); fp2_mul(&t3, &s, &u);
  fp2_sub(&t3, &t3, &aa); fp2_add(&t3, &t3, &bb); fp2_sub(&t3, &t3, &cc);
  fp2_add(&s, &b->c0, &b->c1); fp2_add(&u, &a->c0, &a->c1); fp2_mul(&t2, &s, &u);
  fp2_sub(&t2, &t2, &aa); fp2_sub(&t2, &t2, &bb); fp2_mul_xi(&s, &cc); fp2_add(&t2, &t2, &s);
  r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static void fp6_sqr(fp6* r, const fp6* a) {                        /* :270-293 */
  fp2 s0, ab, s1, s2, bc, s3, s4, t;
  fp2_sqr(&s0, &a->c0); fp2_mul(&ab, &a->c0, &a->c1); fp2_dbl(&s1, &ab);
  fp2_sub(&t, &a->c0, &a->c1); fp2_add(&t, &t, &a->c2); fp2_sqr(&s2, &t);
  fp2_mul(&bc, &a->c1, &a->c2); fp2_dbl(&s3, &bc); fp2_sqr(&s4, &a->c2);
  fp2 c0, c1, c2;
  fp2_mul_xi(&c0, &s3); fp2_add(&c0, &c0, &s0);
  fp2_mul_xi(&c1, &s4); fp2_add(&c1, &c1, &s1);
  fp2_add(&c2, &s1, &s2); fp2_add(&c2, &c2, &s3); fp2_sub(&c2, &c2, &s0); fp2_sub(&c2, &c2, &s4);
  r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fp6_mul_by_01(fp6* r, const fp6* s, const fp2* c0, const fp2* c1) {   /* :232-259 */
  fp2 aa, bb, t1, t2, t3, u, w;
  fp2_mul(&aa, &s->c0, c0); fp2_mul(&bb, &s->c1, c1);
  fp2_mul(&t1, &s->c2, c1); fp2_mul_xi(&t1, &t1); fp2_add(&t1, &t1, &aa);
  fp2_add(&u, c0, c1); fp2_add(&w, &s->c0, &s->c1); fp2_mul(&t2, &u, &w); fp2_sub(&t2, &t2, &aa); fp2_sub(&t2, &t2, &bb);
  fp2_mul(&t3, &s->c2, c0); fp2_add(&t3, &t3, &bb);
  r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static void fp6_mul_by_1(fp6* r, const fp6* s, const fp2* c1) {    /* :261-268 */
  fp2 t0, t1, t2; fp2_mul(&t0, &s->c2, c1); fp2_mul_xi(&t0, &t0); fp2_mul(&t1, &s->c0, c1); fp2_mul(&t2, &s->c1, c1);
  r->c0 = t0; r->c1 = t1; r->c2 = t2;
}
static int fp6_inv(fp6* r, const fp6* a) {                         /* :59-89 */
  fp2 c0, c1, c2, t, u;
  fp2_mul(&t, &a->c1, &a->c2); fp2_mul_xi(&t, &t); fp2_sqr(&c0, &a->c0); fp2_sub(&c0, &c0, &t);
  fp2_sqr(&c1, &a->c2); fp2_mul_xi(&c1, &c1); fp2_mul(&t, &a->c0, &a->c1); fp2_sub(&c1, &c1, &t);
  fp2_sqr(&c2, &a->c1); fp2_mul(&t, &a->c0, &a->c2); fp2_sub(&c2, &c2, &t);
  fp2_mul(&t, &a->c1, &c2); fp2_mul(&u, &a->c2, &c1); fp2_add(&t, &t, &u); fp2_mul_xi(&t, &t);
  fp2_mul(&u, &a->c0, &c0); fp2_add(&t, &t, &u);
  fp2 ti; if (!fp2_inv(&ti, &t)) { memset(r, 0, sizeof *r); return 0; }
  fp2_mul(&r->c0, &c0, &ti); fp2_mul(&r->c1, &c1, &ti); fp2_mul(&r->c2, &c2, &ti); return 1;
}

/* Frobenius constants: canonical values derived at start-up from xi = 1 + u (the reference stores
 * Montgomery-form limbs at fq6_target_tree.rs:139-162, fq12_target_tree.rs:101-119; KAT-4 checks them) */
static fp2 FROB6_C1, FROB6_C2, FROB12_C1;
static fp12 FP12_ONE_C;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void fp2_pow_big(fp2* r, const fp2* a, const uint64_t* e, int nwords) {
  fp2 acc; memset(&acc, 0, sizeof acc); acc.c0 = R1;
  for (int i = nwords * 64 - 1; i >= 0; --i) { fp2_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fp2_mul(&acc, &acc, a); }
  *r = acc;
}
/* q = (a - 1) / d and 2q for multi-word a (d = 3 or 6), simple long division */
static void div_small(uint64_t* q, const uint64_t* a, uint64_t d, int n) {
  u128 rem = 0; for (int i = n - 1; i >= 0; --i) { u128 cur = (rem << 64) | a[i]; q[i] = (uint64_t)(cur / d); rem = cur % d; }
}
static void init_consts(void) {
  fp2 xi; xi.c0 = R1; xi.c1 = R1;
  uint64_t pm1[6], e[6]; memcpy(pm1, P.l, 48); pm1[0] -= 1;
  div_small(e, pm1, 3, 6); fp2_pow_big(&FROB6_C1, &xi, e, 6);           /* xi^((p-1)/3) */
  fp2_sqr(&FROB6_C2, &FROB6_C1);                                          /* xi^((2p-2)/3) */
  div_small(e, pm1, 6, 6); fp2_pow_big(&FROB12_C1, &xi, e, 6);          /* xi^((p-1)/6) */
  memset(&FP12_ONE_C, 0, sizeof FP12_ONE_C); FP12_ONE_C.c0.c0.c0 = R1;
}
static void fp6_frobenius(fp6* r, const fp6* a) {                  /* :129-169 */
  fp2 t; fp2_conj(&r->c0, &a->c0); fp2_conj(&t, &a->c1); fp2_mul(&r->c1, &t, &FROB6_C1); fp2_conj(&t, &a->c2); fp2_mul(&r->c2, &t, &FROB6_C2);
}

/* ------------------------------------------------------------------------------------------------ Fp12
 * src/fields_as_trees/fq12_target_tree.rs:53-176 */
static void fp12_mul(fp12* r, const fp12* a, const fp12* b) {      /* :130-141 */
  fp6 aa, bb, o, s, c1, c0;
  fp6_mul(&aa, &a->c0, &b->c0); fp6_mul(&bb, &a->c1, &b->c1);
  fp6_add(&o, &b->c0, &b->c1); fp6_add(&s, &a->c1, &a->c0); fp6_mul(&c1, &s, &o);
  fp6_sub(&c1, &c1, &aa); fp6_sub(&c1, &c1, &bb);
  fp6_mul_v(&c0, &bb); fp6_add(&c0, &c0, &aa);
  r->c0 = c0; r->c1 = c1;
}
static void fp12_sqr(fp12* r, const fp12* a) {                     /* :143-155 */
  fp6 ab, s, t, c0, c1;
  fp6_mul(&ab, &a->c0, &a->c1); fp6_add(&s, &a->c0, &a->c1);
  fp6_mul_v(&t, &a->c1); fp6_add(&t, &t, &a->c0); fp6_mul(&c0, &t, &s); fp6_sub(&c0, &c0, &ab);
  fp6_add(&c1, &ab, &ab); fp6_mul_v(&t, &ab); fp6_sub(&c0, &c0, &t);
  r->c0 = c0; r->c1 = c1;
}
static void fp12_conj(fp12* r, const fp12* a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }   /* :53-58 */
static int fp12_inv(fp12* r, const fp12* a) {                      /* :77-90 */
  fp6 t0, t1, ti;
  fp6_sqr(&t0, &a->c0); fp6_sqr(&t1, &a->c1); fp6_mul_v(&t1, &t1); fp6_sub(&t0, &t0, &t1);
  if (!fp6_inv(&ti, &t0)) { memset(r, 0, sizeof *r); return 0; }
  fp6_mul(&r->c0, &a->c0, &ti); fp6_neg(&ti, &ti); fp6_mul(&r->c1, &a->c1, &ti); return 1;
}
static void fp12_frobenius(fp12* r, const fp12* a) {               /* :92-128 */
  fp6 c0, c1; fp6_frobenius(&c0, &a->c0); fp6_frobenius(&c1, &a->c1);
  fp2_mul(&c1.c0, &c1.c0, &FROB12_C1); fp2_mul(&c1.c1, &c1.c1, &FROB12_C1); fp2_mul(&c1.c2, &c1.c2, &FROB12_C1);
  r->c0 = c0; r->c1 = c1;
}
static void fp12_mul_by_014(fp12* r, const fp12* f, const fp2* c0, const fp2* c1, const fp2* c4) {   /* :157-176 */
  fp6 aa, bb, s, x, t; fp2 o;
  fp6_mul_by_01(&aa, &f->c0, c0, c1); fp6_mul_by_1(&bb, &f->c1, c4); fp2_add(&o, c1, c4);
  fp6_add(&s, &f->c1, &f->c0); fp6_mul_by_01(&x, &s, c0, &o); fp6_sub(&x, &x, &aa); fp6_sub(&x, &x, &bb);
  fp6_mul_v(&t, &bb); fp6_add(&t, &t, &aa);
  r->c0 = t; r->c1 = x;
}
static void fp12_from_fp2(fp12* r, const fp2* v) { memset(r, 0, sizeof *r); r->c0.c0 = *v; }   /* opt.rs:18-36 */

/* ------------------------------------------------------------------------------------------------ G2 Jacobian
 * ark-ec 0.4.2 [third party] formulas; call sites opt.rs:93, :98 */
typedef struct { fp2 x, y, z; } g2j;
typedef struct { fp x, y, z; } g1j;
static void g2_double(g2j* r, const g2j* p) {
  if (fp2_is_zero(&p->z)) { *r = *p; return; }
  fp2 A, B, C, D, E, F, t, X3, Y3, Z3;
  fp2_sqr(&A, &p->x); fp2_sqr(&B, &p->y); fp2_sqr(&C, &B);
  fp2_add(&t, &p->x, &B); fp2_sqr(&t, &t); fp2_sub(&t, &t, &A); fp2_sub(&t, &t, &C); fp2_dbl(&D, &t);
  fp2_add(&E, &A, &A); fp2_add(&E, &E, &A); fp2_sqr(&F, &E);
  fp2_mul(&Z3, &p->y, &p->z); fp2_dbl(&Z3, &Z3);
  fp2_sub(&X3, &F, &D); fp2_sub(&X3, &X3, &D);
  fp2_dbl(&C, &C); fp2_dbl(&C, &C); fp2_dbl(&C, &C);
  fp2_sub(&t, &D, &X3); fp2_mul(&Y3, &E, &t); fp2_sub(&Y3, &Y3, &C);
  r->x = X3; r->y = Y3; r->z = Z3;
}
static void g2_add(g2j* r, const g2j* p1, const g2j* p2) {
  if (fp2_is_zero(&p1->z)) { *r = *p2; return; }
  if (fp2_is_zero(&p2->z)) { *r = *p1; return; }
  fp2 Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
  fp2_sqr(&Z1Z1, &p1->z); fp2_sqr(&Z2Z2, &p2->z);
  fp2_mul(&U1, &p1->x, &Z2Z2); fp2_mul(&U2, &p2->x, &Z1Z1);
  fp2_mul(&S1, &p1->y, &p2->z); fp2_mul(&S1, &S1, &Z2Z2);
  fp2_mul(&S2, &p2->y, &p1->z); fp2_mul(&S2, &S2, &Z1Z1);
  if (fp2_eq(&U1, &U2) && fp2_eq(&S1, &S2)) { g2_double(r, p1); return; }
  fp2_sub(&H, &U2, &U1); fp2_dbl(&I, &H); fp2_sqr(&I, &I); fp2_mul(&J, &H, &I);
  fp2_sub(&rr, &S2, &S1); fp2_dbl(&rr, &rr); fp2_mul(&V, &U1, &I);
  fp2_sqr(&X3, &rr); fp2_sub(&X3, &X3, &J); fp2_dbl(&t, &V); fp2_sub(&X3, &X3, &t);
  fp2_sub(&t, &V, &X3); fp2_mul(&Y3, &rr, &t); fp2_mul(&t, &S1, &J); fp2_dbl(&t, &t); fp2_sub(&Y3, &Y3, &t);
  fp2_add(&Z3, &p1->z, &p2->z); fp2_sqr(&Z3, &Z3); fp2_sub(&Z3, &Z3, &Z1Z1); fp2_sub(&Z3, &Z3, &Z2Z2); fp2_mul(&Z3, &Z3, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* ================================================================================================ MODE A
 * src/miller_loop_native_optimized.rs */
static void line_function(fp2* n, fp2* d, const g2j* q1, const g2j* q2, const g1j* p) {   /* :8-78 */
  fp2 num, den, t, u, a, b, three, two;
  fp2_mul(&num, &q2->y, &q1->z); fp2_mul(&t, &q1->y, &q2->z); fp2_sub(&num, &num, &t);       /* :14 */
  fp2_mul(&den, &q2->x, &q1->z); fp2_mul(&t, &q1->x, &q2->z); fp2_sub(&den, &den, &t);       /* :15 */
  int vertical = 0;
  if (!fp2_is_zero(&den)) { /* chord :16 */ }
  else if (fp2_is_zero(&num)) {                                                              /* :38 */
    memset(&three, 0, sizeof three); memset(&two, 0, sizeof two);
    fp_set_small(&three.c0, 3); fp_set_small(&two.c0, 2);
    fp2_mul(&num, &three, &q1->x); fp2_mul(&num, &num, &q1->x);                              /* :39 */
    fp2_mul(&den, &two, &q1->y); fp2_mul(&den, &den, &q1->z);                                /* :40 */
  } else vertical = 1;                                                                       /* :62 */
  fp2_mul_fp(&a, &q1->z, &p->x); fp2_mul_fp(&t, &q1->x, &p->z); fp2_sub(&a, &a, &t);        /* xp z1 - x1 zp */
  if (vertical) { *n = a; fp2_mul_fp(d, &q1->z, &p->z); return; }                           /* :66, :73 */
  fp2_mul_fp(&b, &q1->z, &p->y); fp2_mul_fp(&t, &q1->y, &p->z); fp2_sub(&b, &b, &t);        /* yp z1 - y1 zp */
  fp2_mul(&t, &num, &a); fp2_mul(&u, &den, &b); fp2_sub(n, &t, &u);                          /* :20-23 */
  fp2_mul_fp(&t, &den, &p->z); fp2_mul(d, &t, &q1->z);                                       /* :31 */
}

/* optimized_miller_loop, :81-127.  Multiplies full Fq12 values like arkworks does (no Fp2 shortcut). */
static uint8_t miller_ref_opt(fp12* out, const g1j* p, const g2j* q) {
  g2j r = *q; fp12 fn = FP12_ONE_C, fd = FP12_ONE_C, e; fp2 n, d;
  for (int i = 0; i < 64; ++i) {                                     /* :89 LSB first */
    line_function(&n, &d, &r, &r, p);                                /* :90 */
    fp12_mul(&fn, &fn, &fn); fp12_from_fp2(&e, &n); fp12_mul(&fn, &fn, &e);   /* :91 */
    fp12_mul(&fd, &fd, &fd); fp12_from_fp2(&e, &d); fp12_mul(&fd, &fd, &e);   /* :92 */
    g2_add(&r, &r, &r);                                              /* :93 */
    if ((BLS_X >> i) & 1) {
      line_function(&n, &d, &r, q, p);                               /* :95 */
      fp12_from_fp2(&e, &n); fp12_mul(&fn, &fn, &e);                 /* :96 */
      fp12_from_fp2(&e, &d); fp12_mul(&fd, &fd, &e);                 /* :97 */
      g2_add(&r, &r, q);                                             /* :98 */
    }
  }
  fp12 di;
  if (!fp12_inv(&di, &fd)) { memset(out, 0, sizeof *out); return 1; }   /* :102 panics in the reference */
  fp12_mul(&fn, &fn, &di);                                           /* :102 */
  fp12_mul(out, &fn, &fn);                                           /* :116-121 */
  return 0;
}

/* ================================================================================================ MODE B
 * src/miller_loop_native.rs */
typedef struct { fp2 x, y; } g2a;
typedef struct { fp x, y; } g1a;
static void doubling_step(fp2 c[3], g2j* r) {                        /* :27-55 */
  fp2 t0, t1, t2, t3, t4, t5, t6, zs, X3, Y3, Z3, t;
  fp2_sqr(&t0, &r->x); fp2_sqr(&t1, &r->y); fp2_sqr(&t2, &t1);
  fp2_add(&t3, &t1, &r->x); fp2_sqr(&t3, &t3); fp2_sub(&t3, &t3, &t0); fp2_sub(&t3, &t3, &t2); fp2_dbl(&t3, &t3);
  fp2_add(&t4, &t0, &t0); fp2_add(&t4, &t4, &t0); fp2_add(&t6, &r->x, &t4); fp2_sqr(&t5, &t4); fp2_sqr(&zs, &r->z);
  fp2_sub(&X3, &t5, &t3); fp2_sub(&X3, &X3, &t3);
  fp2_add(&Z3, &r->z, &r->y); fp2_sqr(&Z3, &Z3); fp2_sub(&Z3, &Z3, &t1); fp2_sub(&Z3, &Z3, &zs);
  fp2_sub(&t, &t3, &X3); fp2_mul(&Y3, &t, &t4);
  fp2_dbl(&t2, &t2); fp2_dbl(&t2, &t2); fp2_dbl(&t2, &t2); fp2_sub(&Y3, &Y3, &t2);
  fp2_mul(&t3, &t4, &zs); fp2_dbl(&t3, &t3); fp2_neg(&t3, &t3);       /* :44-46 */
  fp2_sqr(&t6, &t6); fp2_sub(&t6, &t6, &t0); fp2_sub(&t6, &t6, &t5);
  fp2_dbl(&t1, &t1); fp2_dbl(&t1, &t1); fp2_sub(&t6, &t6, &t1);
  fp2_mul(&t0, &Z3, &zs); fp2_dbl(&t0, &t0);
  r->x = X3; r->y = Y3; r->z = Z3; c[0] = t0; c[1] = t3; c[2] = t6;
}
static void addition_step(fp2 c[3], g2j* r, const g2a* q) {          /* :58-87 */
  fp2 zs, ys, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, X3, Y3, Z3, t;
  fp2_sqr(&zs, &r->z); fp2_sqr(&ys, &q->y); fp2_mul(&t0, &zs, &q->x);
  fp2_add(&t1, &q->y, &r->z); fp2_sqr(&t1, &t1); fp2_sub(&t1, &t1, &ys); fp2_sub(&t1, &t1, &zs); fp2_mul(&t1, &t1, &zs);
  fp2_sub(&t2, &t0, &r->x); fp2_sqr(&t3, &t2); fp2_dbl(&t4, &t3); fp2_dbl(&t4, &t4); fp2_mul(&t5, &t4, &t2);
  fp2_sub(&t6, &t1, &r->y); fp2_sub(&t6, &t6, &r->y); fp2_mul(&t9, &t6, &q->x); fp2_mul(&t7, &t4, &r->x);
  fp2_sqr(&X3, &t6); fp2_sub(&X3, &X3, &t5); fp2_sub(&X3, &X3, &t7); fp2_sub(&X3, &X3, &t7);
  fp2_add(&Z3, &r->z, &t2); fp2_sqr(&Z3, &Z3); fp2_sub(&Z3, &Z3, &zs); fp2_sub(&Z3, &Z3, &t3);
  fp2_add(&t10, &q->y, &Z3); fp2_sub(&t, &t7, &X3); fp2_mul(&t8, &t, &t6);
  fp2_mul(&t0, &r->y, &t5); fp2_dbl(&t0, &t0); fp2_sub(&Y3, &t8, &t0);
  fp2_sqr(&t10, &t10); fp2_sub(&t10, &t10, &ys); fp2_sqr(&t, &Z3); fp2_sub(&t10, &t10, &t);
  fp2_dbl(&t9, &t9); fp2_sub(&t9, &t9, &t10); fp2_dbl(&t10, &Z3);
  fp2_neg(&t6, &t6); fp2_dbl(&t1, &t6);                               /* :83-84 */
  r->x = X3; r->y = Y3; r->z = Z3; c[0] = t10; c[1] = t1; c[2] = t9;
}
static void ell(fp12* f, const fp2 c[3], const g1a* p) {             /* commented :139-152 */
  fp2 c0, c1; fp2_mul_fp(&c0, &c[0], &p->y); fp2_mul_fp(&c1, &c[1], &p->x);
  fp12_mul_by_014(f, f, &c[2], &c1, &c0);
}
#define ORACLE_MAX_K 64   /* pairs per product (the reference takes a slice of any length, src/miller_loop_native.rs:154) */
static void multi_miller(fp12* out, const g1a* p, const g2a* q, const int* skip, int k) {   /* :89-116, :154-212 */
  g2j r[ORACLE_MAX_K]; fp2 c[3]; fp12 f = FP12_ONE_C;
  for (int j = 0; j < k; ++j) { r[j].x = q[j].x; r[j].y = q[j].y; memset(&r[j].z, 0, sizeof(fp2)); r[j].z.c0 = R1; }
  for (int b = 61; b >= 0; --b) {
    int bit = ((BLS_X >> 1) >> b) & 1;
    for (int j = 0; j < k; ++j) { doubling_step(c, &r[j]); if (!skip[j]) ell(&f, c, &p[j]); }
    if (bit) for (int j = 0; j < k; ++j) { addition_step(c, &r[j], &q[j]); if (!skip[j]) ell(&f, c, &p[j]); }
    fp12_sqr(&f, &f);
  }
  for (int j = 0; j < k; ++j) { doubling_step(c, &r[j]); if (!skip[j]) ell(&f, c, &p[j]); }
  fp12_conj(out, &f);
}
/* src/fields_as_trees/miller_loop.rs:29-126 */
static void fp4_square(fp2* c0, fp2* c1, const fp2* a, const fp2* b) {
  fp2 t0, t1, t2; fp2_sqr(&t0, a); fp2_sqr(&t1, b); fp2_mul_xi(&t2, &t1); fp2_add(c0, &t2, &t0);
  fp2_add(&t2, a, b); fp2_sqr(&t2, &t2); fp2_sub(&t2, &t2, &t0); fp2_sub(c1, &t2, &t1);
}
static void cyclotomic_square(fp12* r, const fp12* f) {
  fp2 z0 = f->c0.c0, z4 = f->c0.c1, z3 = f->c0.c2, z2 = f->c1.c0, z1 = f->c1.c1, z5 = f->c1.c2, t0, t1, t2, t3;
  fp4_square(&t0, &t1, &z0, &z1);
  fp2_sub(&z0, &t0, &z0); fp2_dbl(&z0, &z0); fp2_add(&z0, &z0, &t0);
  fp2_add(&z1, &t1, &z1); fp2_dbl(&z1, &z1); fp2_add(&z1, &z1, &t1);
  fp4_square(&t0, &t1, &z2, &z3); fp4_square(&t2, &t3, &z4, &z5);
  fp2_sub(&z4, &t0, &z4); fp2_dbl(&z4, &z4); fp2_add(&z4, &z4, &t0);
  fp2_add(&z5, &t1, &z5); fp2_dbl(&z5, &z5); fp2_add(&z5, &z5, &t1);
  fp2_mul_xi(&t0, &t3); fp2_add(&z2, &t0, &z2); fp2_dbl(&z2, &z2); fp2_add(&z2, &z2, &t0);
  fp2_sub(&z3, &t2, &z3); fp2_dbl(&z3, &z3); fp2_add(&z3, &z3, &t2);
  r->c0.c0 = z0; r->c0.c1 = z4; r->c0.c2 = z3; r->c1.c0 = z2; r->c1.c1 = z1; r->c1.c2 = z5;
}
static void cyclotomic_exp(fp12* r, const fp12* f) {                 /* :106-126, product at :121 kept */
  fp12 tmp = FP12_ONE_C; int found = 0;
  for (int b = 63; b >= 0; --b) {
    int bit = (BLS_X >> b) & 1;
    if (found) cyclotomic_square(&tmp, &tmp); else found = bit;
    if (bit) fp12_mul(&tmp, &tmp, f);
  }
  fp12_conj(r, &tmp);
}
static int final_exponentiation(fp12* out, const fp12* f) {          /* :166-178, :128-164 */
  fp12 t0, t1, t2, t3, t4, t5, t6, x;
  fp12_conj(&t0, f);
  if (!fp12_inv(&t1, f)) { memset(out, 0, sizeof *out); return 0; }
  fp12_mul(&t2, &t0, &t1); t1 = t2;
  fp12_frobenius(&t2, &t2); fp12_frobenius(&t2, &t2); fp12_mul(&t2, &t2, &t1);
  cyclotomic_square(&t1, &t2); fp12_conj(&t1, &t1);
  cyclotomic_exp(&t3, &t2); cyclotomic_square(&t4, &t3); fp12_mul(&t5, &t1, &t3);
  cyclotomic_exp(&t1, &t5); cyclotomic_exp(&t0, &t1); cyclotomic_exp(&t6, &t0);
  fp12_mul(&t6, &t6, &t4); cyclotomic_exp(&t4, &t6); fp12_conj(&t5, &t5);
  fp12_mul(&x, &t5, &t2); fp12_mul(&t4, &t4, &x); fp12_conj(&t5, &t2);
  fp12_mul(&t1, &t1, &t2); fp12_frobenius(&t1, &t1); fp12_frobenius(&t1, &t1); fp12_frobenius(&t1, &t1);
  fp12_mul(&t6, &t6, &t5); fp12_frobenius(&t6, &t6);
  fp12_mul(&t3, &t3, &t0); fp12_frobenius(&t3, &t3); fp12_frobenius(&t3, &t3);
  fp12_mul(&t3, &t3, &t1); fp12_mul(&t3, &t3, &t6); fp12_mul(out, &t3, &t4);
  return 1;
}

/* ================================================================================================ batch API
 * Buffers as in include/bls381_hip.h, arkworks Montgomery form only (6 x u64 per Fp). */
static int all_zero(const uint64_t* w, int n) { uint64_t o = 0; for (int i = 0; i < n; ++i) o |= w[i]; return o == 0; }

typedef struct { int kind; const uint64_t *a, *b; uint64_t* out; uint8_t* flags; size_t lo, hi, k; } job_t;

static void run_range(const job_t* j) {
  for (size_t i = j->lo; i < j->hi; ++i) {
    fp12 f; uint8_t fl = 0;
    if (j->kind == 0) {
      g1j p; g2j q; memcpy(&p, j->a + 18 * i, 144); memcpy(&q, j->b + 36 * i, 288);
      fl = miller_ref_opt(&f, &p, &q) ? 1 : 0;
    } else if (j->kind == 1 || j->kind == 2) {
      g1a p[ORACLE_MAX_K]; g2a q[ORACLE_MAX_K]; int skip[ORACLE_MAX_K]; size_t k = j->k;
      for (size_t t = 0; t < k; ++t) {
        const uint64_t* pa = j->a + 12 * (k * i + t); const uint64_t* qa = j->b + 24 * (k * i + t);
        memcpy(&p[t], pa, 96); memcpy(&q[t], qa, 192);
        skip[t] = all_zero(pa, 12) || all_zero(qa, 24); if (skip[t]) fl |= 2;
      }
      multi_miller(&f, p, q, skip, (int)k);
      if (j->kind == 2) { fp12 g; final_exponentiation(&g, &f); f = g; }
    } else {
      fp12 g; memcpy(&g, j->a + 72 * i, 576); final_exponentiation(&f, &g);
    }
    memcpy(j->out + 72 * i, &f, 576);
    if (j->flags) j->flags[i] = fl;
  }
}
static void* worker(void* arg) { run_range((const job_t*)arg); return NULL; }

static int run_batch(int kind, const uint64_t* a, const uint64_t* b, size_t k, uint64_t* out, uint8_t* flags, size_t n, int threads) {
  pthread_once(&g_once, init_consts);
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = (job_t){kind, a, b, out, flags, n * t / threads, n * (t + 1) / threads, k};
    if (t) pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  run_range(&jobs[0]);
  for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}

int oracle_miller_ref_opt_batch(const uint64_t* g1_jac, const uint64_t* g2_jac, uint64_t* out, uint8_t* flags, size_t n, int threads) {
  return run_batch(0, g1_jac, g2_jac, 1, out, flags, n, threads);
}
int oracle_multi_miller_batch(const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k, uint64_t* out, uint8_t* flags, size_t n, int threads) {
  if (k < 1 || k > ORACLE_MAX_K) return 1;
  return run_batch(1, g1_aff, g2_aff, k, out, flags, n, threads);
}
int oracle_multi_pairing_batch(const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k, uint64_t* out, uint8_t* flags, size_t n, int threads) {
  if (k < 1 || k > ORACLE_MAX_K) return 1;
  return run_batch(2, g1_aff, g2_aff, k, out, flags, n, threads);
}
int oracle_final_exp_batch(const uint64_t* in, uint64_t* out, size_t n, int threads) {
  return run_batch(3, in, NULL, 1, out, NULL, n, threads);
}
/* single-op surfaces for cross-checking against pyref: op 0 mul, 1 sqr, 2 inv, 3 frobenius, 4 cyclotomic square */
int oracle_fp12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  pthread_once(&g_once, init_consts);
  for (size_t i = 0; i < n; ++i) {
    fp12 x, y, r; memcpy(&x, a + 72 * i, 576); if (b) memcpy(&y, b + 72 * i, 576); else y = x;
    switch (op) {
      case 0: fp12_mul(&r, &x, &y); break;
      case 1: fp12_sqr(&r, &x); break;
      case 2: fp12_inv(&r, &x); break;
      case 3: fp12_frobenius(&r, &x); break;
      case 4: cyclotomic_square(&r, &x); break;
      default: return 1;
    }
    memcpy(out + 72 * i, &r, 576);
  }
  return 0;
}
int oracle_fp_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {   /* 0 add 1 sub 2 mul 4 inv */
  for (size_t i = 0; i < n; ++i) {
    fp x, y, r; memcpy(&x, a + 6 * i, 48); if (b) memcpy(&y, b + 6 * i, 48); else y = x;
    switch (op) {
      case 0: fp_add(&r, &x, &y); break;
      case 1: fp_sub(&r, &x, &y); break;
      case 2: fp_mul(&r, &x, &y); break;
      case 4: fp_inv(&r, &x); break;
      default: return 1;
    }
    memcpy(out + 6 * i, &r, 48);
  }
  return 0;
}
