// Mutation testing of the march's exactness machinery (round 5; tools/mutants.py, profiles/r05_mutants.md).
//
// The march skips samples on the strength of a dozen hand-derived safety margins, and every skipped sample is a claim about
// the reference's minimum over ALL samples (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:510-514).  -DGCFR_MUT=<n>
// builds the library with ONE of those margins removed or inverted; the -m gpu suite must then fail.  A margin whose removal
// no test notices is a margin nobody is testing (round 3 shipped a hole that 3.4 G random soak pixels had not found).
//
//     GCFR_M(n, mutant tokens, product tokens)
//
// expands to the product tokens unless GCFR_MUT == n: pure token selection by the preprocessor, so the product build
// (GCFR_MUT undefined) compiles exactly the token stream it had before the mutants were written -- tools/compare_device_code.py
// against the build without them: every kernel identical.
//
//  n  margin (file: where)                                                          mutant
//  1  mask bounding box inflated by 0.5 (rint) + 0.01 (gcfr_march.hpp, candidate range)   0.51 -> 0.49
//  2  bounding octagon inflated by 1 + 0.02                                          1.02 -> 0.98
//  3  Kerr: K1 (roundings proportional to |BCz|, |c1| t)                            = 0
//  4  Kerr: K2 r (roundings proportional to |BA|)                                    = 0
//  5  Kerr: plane-evaluation term n (1.2e-2 + 8e-6 max(H, W))                        = 0
//  6  depth-bound skip: bound^2 * 0.998 > running minimum                            0.998 -> 1.002
//  7  early termination, main loop: the same slack                                   0.998 -> 1.002
//  8  early termination, trailing loop (horizon cap): the same slack                 0.998 -> 1.002
//  9  safeS = 0.98e12 den^2 ("the distance is certainly below the masked 1e6")       0.98e12 -> 1.02e12
// 10  gap0: the isolated sampled value z = 0 at integral coordinates                 dropped from the bound
// 11  c1 > 0 ("the ray is still rising") in both termination tests                   dropped
// 13  horizon tables: wrap partners (prefix tables include the last column / row, suffix the first)   left out
// 14  horizon tables: the dilation of the live cells wraps where the gathers do      no wrap
//     (round 6, on 13 / 14 and the audit build, profiles/r06_audit_wrap.json.  tests/margin_scenes.py family `wrap_column` -- a
//      wall that ramps with the rays' own height in the masked-out wrap-partner column / row, light point exactly on the image's
//      edge -- makes mutant 14 contradict 51,164 termination claims and fail tests/test_gpu_margins.py end to end.  Mutant 13
//      changes nothing there, and with the reference's sample tables (t <= 0.82, T8:468) it cannot: a sample reads column -1
//      (= W-1) only if floor(u_x) = -1, i.e. x + t dx < -W/2 + 1e-4, and the end point is clamped to x >= -W/2 (T8:462-465); a lane
//      heading LEFT (dx < 0: the prefix table) starts at x >= -W/2 + 1 and needs t > 1 - 1e-4 / |dx| to get there, a lane in column
//      0 has dx >= 0 and looks up the SUFFIX table, which holds column W-1 by construction; rows likewise.  A caller's table that
//      reaches t = 1 (the prepass accepts tables inside [0, 1]) does let a prefix lane's LAST sample read the wrap partner: family
//      `wrap_last_sample` marches such a table over a wall sized so that this last sample is the minimum -- mutant 13 then
//      contradicts 242,244 termination claims (14: 261,834) and fails test_gpu_margins.py[wrap_last_sample-*]; the product: 0.)
// 15  march_grid: a tile that hands itself to the rough variant is re-run by it      not re-run
// 16  tie predecessor (first index of the minimal DISTANCE, torch.min)               never re-marched
// 17  pixels = mask: a pixel whose own mask cell is zero is not marched (value 1e6)   marched after all
// 18  trailing loop entered / continued only when bestS < safeS                      condition dropped
// 20  horizon tables: live cells = the mask's non-zero cells dilated by one          no dilation
// 21  horizon tables: an entry covers two cells behind / three ahead of its own       none (gcfr_shadow.hip)
// 22  bounds grid stride: a group's footprint + 3 cells                              + 1
// 23  Kerr as a whole                                                                = 0
// 24  candidate range pruned some sample of the wave's range -> any_masked           not recorded
// 25  termination cap without horizon tables: max(image depth maximum, 0)            without the 0
// 26  "every remaining sample of the lane lies outside the mask's box" (lane_last < next group's first sample)   off by one
// 27  trailing loop: "the lane has left the box" (lane_last < this group's first sample)   off by one
// 29  bound evaluated at the group's first AND last sample (linear in t)             first sample only
//
// 3 and 5 are the two mutants no END-TO-END test kills (profiles/r05_mutants.md): K1, K2 r and the plane term budget for three different
// effects -- the 1e-4 position offset times |BCz|, the f32 roundings of the distance's products, the offset times the surface's slope --
// each ten times over, and they are ADDED: with one of them gone the other two and the 0.2 % slack still cover its effect wherever a bound
// is tight enough to DECIDE a minimum (K1's effect exceeds the rest only within 43 px of an overhead light's foot, where the ray climbs
// 4000 t and no bounds tile is a thin band; the plane term's only on slopes near the plane fit's limit of 4, where the depth range
// inflates K2 r).  Kerr as a whole (23) dies in five scene families, K2 r alone (4) far from zero.  The AUDIT build kills both
// (-DGCFR_COUNTERS -DGCFR_AUDIT, gcfr_march.hpp; tools/audit.py, tests/test_gpu_audit.py): it checks the claim the margins were derived
// for -- g > 0 => S_k >= 0.998 g^2 for every unmasked sample of the group -- at every evaluation of the bound, decisive or not, and
// without K1 (on `pits2`) or without the plane term (on `facets`) hundreds of evaluations contradict it.
//
// NOT in the list, because removing them cannot change a result (round 5 built them, they survived, and the reason is a proof, not
// a missing test):
//  * candidate range, "one sample of slack either side" (floor(ka) - 1, ceil(kb) + 1): floor / ceil already err on the safe side by
//    up to a whole step -- the first EXCLUDED sample floor(ka) - 1 sits at least one mean step in front of t_a in the uniform model,
//    and an accepted table deviates from that model by < 0.08 steps (every interval within 0.1 % of the mean, 160 of them), the f32
//    index arithmetic by 1e-6: it lies outside the inflated box without the extra sample.  Belt and braces; costs two samples per ray.
//  * trailing loop, `any_masked |= gone`: a lane that is "gone" (past its last candidate sample) when a group is CONSUMED has just had
//    that group's mask bytes read, all zero (outside the mask's box), so any_masked is already set; a lane that goes in the
//    trailing loop was lazy when it entered, i.e. held bestS < safeS, and its any_masked can no longer matter (d < 1e6).
//  * pixels = mask, "a lane outside the image counts as own-pixel-off": such a lane repeats pixel (H-1, W-1) and stores nothing;
//    marched or not, it can only make its wave execute MORE groups.
#pragma once

#ifndef GCFR_MUT
#define GCFR_MUT 0
#endif

#ifndef GCFR_MUT_SLACK_VALUE   // (mutants 7 / 8: what replaces the termination tests' 0.998; tools/mutants.py builds 1.002)
#define GCFR_MUT_SLACK_VALUE 1.002f
#endif
#define GCFR_PP_CAT_(a, b, c) a##b##_##c
#define GCFR_PP_CAT(a, b, c) GCFR_PP_CAT_(a, b, c)
#define GCFR_PP_SECOND_(a, b, ...) b
#define GCFR_PP_SECOND(...) GCFR_PP_SECOND_(__VA_ARGS__)
#define GCFR_PP_IS_PAIR(x) GCFR_PP_SECOND(x, 0, ~)
#define GCFR_PP_IF_0(mut, prod) prod
#define GCFR_PP_IF_1(mut, prod) mut
#define GCFR_PP_IF_(c) GCFR_PP_IF_##c
#define GCFR_PP_IF(c) GCFR_PP_IF_(c)
// GCFR_MUT_PAIR_<GCFR_MUT>_<n> is defined (as a two-element list whose second element is 1) only where both are equal
#define GCFR_M(n, mut, prod) GCFR_PP_IF(GCFR_PP_IS_PAIR(GCFR_PP_CAT(GCFR_MUT_PAIR_, GCFR_MUT, n)))(mut, prod)

#define GCFR_MUT_PAIR_1_1 ~, 1
#define GCFR_MUT_PAIR_2_2 ~, 1
#define GCFR_MUT_PAIR_3_3 ~, 1
#define GCFR_MUT_PAIR_4_4 ~, 1
#define GCFR_MUT_PAIR_5_5 ~, 1
#define GCFR_MUT_PAIR_6_6 ~, 1
#define GCFR_MUT_PAIR_7_7 ~, 1
#define GCFR_MUT_PAIR_8_8 ~, 1
#define GCFR_MUT_PAIR_9_9 ~, 1
#define GCFR_MUT_PAIR_10_10 ~, 1
#define GCFR_MUT_PAIR_11_11 ~, 1
#define GCFR_MUT_PAIR_13_13 ~, 1
#define GCFR_MUT_PAIR_14_14 ~, 1
#define GCFR_MUT_PAIR_15_15 ~, 1
#define GCFR_MUT_PAIR_16_16 ~, 1
#define GCFR_MUT_PAIR_17_17 ~, 1
#define GCFR_MUT_PAIR_18_18 ~, 1
#define GCFR_MUT_PAIR_20_20 ~, 1
#define GCFR_MUT_PAIR_21_21 ~, 1
#define GCFR_MUT_PAIR_22_22 ~, 1
#define GCFR_MUT_PAIR_23_23 ~, 1
#define GCFR_MUT_PAIR_24_24 ~, 1
#define GCFR_MUT_PAIR_25_25 ~, 1
#define GCFR_MUT_PAIR_26_26 ~, 1
#define GCFR_MUT_PAIR_27_27 ~, 1
#define GCFR_MUT_PAIR_29_29 ~, 1
