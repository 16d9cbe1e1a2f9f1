/*
 * skder_amd_spec.h -- the numeric definition of the ANI engine, shared by the HIP product
 * (skder_amd/csrc) and by the CPU oracle (oracle/ani_oracle.c).  Constants only; no code.
 *
 * Origin of each constant: skani's defaults as the reference invokes it (no -c/-m/-k flags are
 * ever passed: /root/reference/src/skDER/skder.py:16-18, 58-59, 103, 119), restated from skani's
 * published description and pinned against the reference's golden tables where they can be
 * (DESIGN.md "Oracle").  skani itself is a third-party, version-unpinned dependency
 * whose source is not under /root/reference.
 */
#ifndef SKDER_AMD_SPEC_H
#define SKDER_AMD_SPEC_H

#define ANI_K              15      /* seed k-mer length */
#define ANI_C              125     /* seed FracMinHash compression: keep if hash < 2^64/125 */
#define ANI_MARKER_K       21      /* marker k-mer length */
#define ANI_MARKER_C       1000    /* marker compression */
#define ANI_MIN_CONTIG     500     /* FASTA records shorter than this are ignored */
#define ANI_CHUNK_LEN      20000   /* the chunked genome is cut into 20 kb windows per contig */
#define ANI_BAND           50      /* chaining look-back, anchors */
#define ANI_BP_BAND        2500    /* chaining look-back, bases on the chunked genome */
#define ANI_MAX_GAP        300     /* max |dq - dr| between chained anchors */
#define ANI_MAX_LIN        5000    /* max dq or dr between chained anchors */
#define ANI_ANCHOR_SCORE   20
#define ANI_MIN_ANCHORS    3       /* shorter chains are dropped */
#define ANI_PAD            250     /* 2*c bases added to every kept chain's span (G5 fit) */
#define ANI_SMALL_PASS     20      /* marker sets smaller than this always pass the screen */
#define ANI_REP_FLOOR      30      /* repetitive-k-mer cut-off is disabled below this multiplicity */
#define ANI_REP_HIST       4096    /* multiplicities are clamped to REP_HIST-1 when ranking them */
#define ANI_REF_OVERLAP_NUM 1      /* a chain is dropped if > NUM/DEN of its span on the other   */
#define ANI_REF_OVERLAP_DEN 2      /* genome is already covered by one better-scoring kept chain */
#define ANI_ROOT_ITERS     24      /* Newton iterations of the fixed-point k-th root */

/* "learned ANI" stand-in: piecewise-linear map on d = 100*(1-ANI_raw), fitted to golden table G5
 * by oracle/fit_calibration.py (rms 0.16, max 0.60 ANI points on 561 pairs); slope 1 beyond the
 * last knot.  UNPINNED outside 96.4 <= ANI <= 100 on one species. */
#define ANI_CAL_N 7
#define ANI_CAL_X {0.00, 0.10, 0.50, 1.00, 1.50, 2.00, 2.50}
#define ANI_CAL_Y {0.0000, 0.1426, 0.6570, 1.4330, 2.1251, 2.7315, 3.2565}

/* output record of one genome pair (device and host layout) */
#define ANI_FX_ONE 4294967296.0    /* fixed-point scale of per-chain ANI estimates: 2^32 */

#endif
