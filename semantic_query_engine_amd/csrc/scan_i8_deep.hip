// scan_i8_deep.hip -- the int8 ping-pong collect scan (scan_i8.hip) built with a FIVE-stage row ring (its variant bit 64) for
// batches of ONE 256-query block.  With one query block per chunk nobody shares a DB tile: every tile comes from HBM to exactly one
// CU, whose bytes in flight (three half-steps of rows = 48 KiB) set the rate -- 0.49 of HBM at batch 256.  The fifth stage issues the
// rows of half-step x in T_{x-4} instead of T_{x-3}: one more period of latency cover at the same LDS footprint class (144 KiB ring).
// Measured, 10 M x 1024 (profiles/r04_search/ab_deep_ring.log): batch 256 2.67 -> 2.56-2.58 ms; batch 1024 (four query blocks share
// each tile through L2) 8.77-8.79 -> 8.82-8.86 ms, so the four-stage build keeps every batch above 256.
#define SQE_I8_VARIANT 65
#define scan_i8_pp_kernel scan_i8_pp_deep_kernel
#define sample_i8_pp_kernel sample_i8_pp_deep_kernel
#define scan_i8_small_kernel scan_i8_small_deep_kernel
#define launch_scan_i8 launch_scan_i8_deep
#define launch_sample_i8 launch_sample_i8_deep
#include "scan_i8.hip"
