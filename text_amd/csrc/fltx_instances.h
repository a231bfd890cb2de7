/*
 * fltx_instances.h -- every decode-kernel instantiation the host can launch
 * (launchDecode in fltx_api.cpp), as FLTX_INST(name<args>) lines grouped so
 * that one group of one workgroup size is one translation unit.
 *
 *   fltx_api.cpp    : no FLTX_INST_W  -> all groups x all sizes (extern template)
 *   fltx_kinst.cpp  : -DFLTX_INST_W=512 -DFLTX_INST_G=1 -> that group only
 *
 * (No include guard: it is included with different FLTX_INST definitions.)
 */
#define FLTX_G1(W) /* lane-per-slot step, 4 tokens per wave */ \
  FLTX_INST(fltx_decode_kernel_lane<W, 4, false, false>)      \
  FLTX_INST(fltx_decode_kernel_lane<W, 4, false, true>)       \
  FLTX_INST(fltx_decode_kernel_lane<W, 4, true, false>)       \
  FLTX_INST(fltx_decode_kernel_lane<W, 4, true, true>)
#define FLTX_G2(W) /* lane-per-slot step, 8 tokens per wave */ \
  FLTX_INST(fltx_decode_kernel_lane<W, 8, false, false>)      \
  FLTX_INST(fltx_decode_kernel_lane<W, 8, false, true>)       \
  FLTX_INST(fltx_decode_kernel_lane<W, 8, true, false>)       \
  FLTX_INST(fltx_decode_kernel_lane<W, 8, true, true>)
#define FLTX_G3(W) /* lean step, groups in registers / streaming */ \
  FLTX_INST(fltx_decode_kernel_lds<W, 6>)                          \
  FLTX_INST(fltx_decode_kernel_lds<W, 12>)                         \
  FLTX_INST(fltx_decode_kernel_lds<W, 255>)
#define FLTX_G4(W) FLTX_INST(fltx_decode_kernel_lds<W, 0>) /* generic engine */
#define FLTX_G5(W) FLTX_INST(fltx_decode_kernel_lds_spec<W, true, true>)
#define FLTX_G6(W) FLTX_INST(fltx_decode_kernel_lds_spec<W, true, false>)
#define FLTX_G7(W) FLTX_INST(fltx_decode_kernel_lds_spec<W, false, true>)
#define FLTX_G8(W) FLTX_INST(fltx_decode_kernel_gws<W>)
#define FLTX_G9(W) FLTX_INST(fltx_decode_kernel_gwslean<W>)
/* lane = LM state decode (fltx_slane.h): (threads, list positions per wave) pairs; W is ignored */
#define FLTX_SLANE_SET(LA, PROF)                               \
  FLTX_INST(fltx_decode_kernel_slane<320, 10, LA, PROF>)       \
  FLTX_INST(fltx_decode_kernel_slane<384, 7, LA, PROF>)        \
  FLTX_INST(fltx_decode_kernel_slane<448, 6, LA, PROF>)        \
  FLTX_INST(fltx_decode_kernel_slane<512, 5, LA, PROF>)        \
  FLTX_INST(fltx_decode_kernel_slane<576, 4, LA, PROF>)        \
  FLTX_INST(fltx_decode_kernel_slane<640, 4, LA, PROF>)        \
  FLTX_INST(fltx_decode_kernel_slane<512, 12, LA, PROF>)       \
  FLTX_INST(fltx_decode_kernel_slane<576, 10, LA, PROF>)
#define FLTX_G10(W) FLTX_SLANE_SET(false, false)
#define FLTX_G11(W) FLTX_SLANE_SET(false, true)
#define FLTX_G15(W) FLTX_SLANE_SET(true, false) /* logAdd */
/* ... with a token-level n-gram LM (TL): the same geometries */
#define FLTX_TLANE_SET(LA)                                     \
  FLTX_INST(fltx_decode_kernel_tlane<320, 10, LA>)             \
  FLTX_INST(fltx_decode_kernel_tlane<384, 7, LA>)              \
  FLTX_INST(fltx_decode_kernel_tlane<448, 6, LA>)              \
  FLTX_INST(fltx_decode_kernel_tlane<512, 5, LA>)              \
  FLTX_INST(fltx_decode_kernel_tlane<576, 4, LA>)              \
  FLTX_INST(fltx_decode_kernel_tlane<640, 4, LA>)              \
  FLTX_INST(fltx_decode_kernel_tlane<512, 12, LA>)             \
  FLTX_INST(fltx_decode_kernel_tlane<576, 10, LA>)
#define FLTX_G28(W) FLTX_TLANE_SET(false)                     \
  FLTX_INST(fltx_decode_kernel_tlane<576, 4, false, true>)    \
  FLTX_INST(fltx_decode_kernel_tlane<512, 5, false, true>) /* (phase clocks: bench.py --profile) */
#define FLTX_G29(W) FLTX_TLANE_SET(true) /* logAdd */
#define FLTX_G30(W) /* stream chunks with a token LM */        \
  FLTX_INST(fltx_decode_kernel_tlane_stream<576, 4>)          \
  FLTX_INST(fltx_decode_kernel_tlane_stream<512, 5>)          \
  FLTX_INST(fltx_decode_kernel_tlane_stream<576, 10>)
#define FLTX_G16(W) /* stream chunks */                     \
  FLTX_INST(fltx_decode_kernel_slane_stream<576, 4>)         \
  FLTX_INST(fltx_decode_kernel_slane_stream<512, 5>)         \
  FLTX_INST(fltx_decode_kernel_slane_stream<576, 10>)
/* ... with several lane groups (fltx_mlane.h): (threads, list positions per wave and group, lane groups, groups per token
 * wave, groups per self wave).  One line here = one row of kMlaneGeo in fltx_api.cpp. */
#define FLTX_MLANE_SET(LA)                                     \
  FLTX_INST(fltx_decode_kernel_mlane<640, 4, 2, 2, 1, LA>)     \
  FLTX_INST(fltx_decode_kernel_mlane<960, 5, 2, 1, 1, LA>)     \
  FLTX_INST(fltx_decode_kernel_mlane<640, 10, 2, 2, 1, LA>)    \
  FLTX_INST(fltx_decode_kernel_mlane<768, 4, 4, 4, 1, LA>)     \
  FLTX_INST(fltx_decode_kernel_mlane<960, 5, 4, 2, 2, LA>)     \
  FLTX_INST(fltx_decode_kernel_mlane<960, 11, 4, 2, 2, LA>)    \
  FLTX_INST(fltx_decode_kernel_mlane<960, 10, 8, 2, 4, LA>)
#define FLTX_G18(W) FLTX_MLANE_SET(false)
#define FLTX_G19(W) FLTX_MLANE_SET(true) /* logAdd */
/* lane = (LM state, trie node) decode (fltx_xlane.h): three waves do not evaluate listed tokens */
#define FLTX_XLANE_SET(HM, PROF)                               \
  FLTX_INST(fltx_decode_kernel_xlane<512, 2, HM, PROF>)        \
  FLTX_INST(fltx_decode_kernel_xlane<640, 2, HM, PROF>)        \
  FLTX_INST(fltx_decode_kernel_xlane<512, 3, HM, PROF>)        \
  FLTX_INST(fltx_decode_kernel_xlane<576, 5, HM, PROF>)        \
  FLTX_INST(fltx_decode_kernel_xlane<640, 10, HM, PROF>)
#define FLTX_G12(W) FLTX_XLANE_SET(0, false) FLTX_XLANE_SET(0, true)
/* ... with logAdd merges (memo in LDS / in HBM) */
#define FLTX_XLANE_LA_SET(HM)                                        \
  FLTX_INST(fltx_decode_kernel_xlane<512, 2, HM, false, true>)        \
  FLTX_INST(fltx_decode_kernel_xlane<640, 2, HM, false, true>)        \
  FLTX_INST(fltx_decode_kernel_xlane<512, 3, HM, false, true>)        \
  FLTX_INST(fltx_decode_kernel_xlane<576, 5, HM, false, true>)        \
  FLTX_INST(fltx_decode_kernel_xlane<640, 10, HM, false, true>)
#define FLTX_G24(W) FLTX_XLANE_LA_SET(0) FLTX_XLANE_LA_SET(1)
#define FLTX_G17(W) FLTX_XLANE_SET(1, false) /* memo in HBM: shares a CU */
/* ... with LM terms, two lane groups (fltx_ylane.h): (threads, groups, rounds, LM terms, memo in HBM = shares a CU) */
#define FLTX_YLANE_SET(PROF)                                  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 0, 0, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 1, 0, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 0, 0, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 1, 0, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 0, 1, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 1, 1, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 0, 1, PROF>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 1, 1, PROF>)
#define FLTX_G13(W) FLTX_YLANE_SET(false)
/* four lane groups (beams 129 .. 256): 1024 threads = ten token waves, four waves for the lanes' own groups, the
 * word wave and the staging wave; LM-state memo in HBM */
#define FLTX_G20(W)                                           \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 0, 1, false>) \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 1, 1, false>) \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 2, 1, false>) \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 3, 1, false>)
/* ASG criterion (LMK bit 1): the geometries of FLTX_YLANE_SET */
#define FLTX_G21(W)                                           \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 2, 0, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 3, 0, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 2, 0, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 3, 0, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 2, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 3, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 2, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 3, 1, false>)
/* spellings shared by several words (LMK bit 2), with the LM terms: memo in HBM (the larger merge table takes its place) */
#define FLTX_G23(W)                                           \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 5, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 7, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 5, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 7, 1, false>)
#define FLTX_G14(W) FLTX_YLANE_SET(true)
/* logAdd merges (LMK bit 3), CTC, with and without the LM terms: the geometries of FLTX_YLANE_SET and FLTX_G20 */
#define FLTX_G25(W)                                            \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 8, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 9, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 8, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 9, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 8, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 9, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 8, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 9, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 8, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 9, 1, false>)
/* ... under ASG (10 / 11) and over spellings with several words (13 / 15: memo in HBM, as FLTX_G23) */
#define FLTX_G26(W)                                             \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 10, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 11, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 10, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 11, 0, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 10, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 11, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 10, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 2, 4, 11, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 10, 1, false>)  \
  FLTX_INST(fltx_decode_kernel_ylane<1024, 4, 4, 11, 1, false>)
#define FLTX_G27(W)                                             \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 13, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<512, 1, 2, 15, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 13, 1, false>)   \
  FLTX_INST(fltx_decode_kernel_ylane<768, 2, 4, 15, 1, false>)
/* lane = LM state decode over a token beam of a large token set (fltx_wlane.h): (threads, list positions per wave) */
#define FLTX_G22(W)                                   \
  FLTX_INST(fltx_decode_kernel_wlane<576, 5>)         \
  FLTX_INST(fltx_decode_kernel_wlane<576, 8>)         \
  FLTX_INST(fltx_decode_kernel_wlane<576, 10>)

/* fltx_mlane.h with a token-level n-gram LM: beams 65 .. 128 / 256 / 512 over token lists of up to 30; the wide rows:
 * beams up to 128 / 256 over lists of up to 64 */
#define FLTX_G31(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 5, 2, 1, 1, false>)
#define FLTX_G32(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 5, 4, 2, 2, false>)
#define FLTX_G33(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 10, 8, 2, 4, false>)
#define FLTX_G34(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 11, 2, 1, 1, false>)
#define FLTX_G35(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 11, 4, 2, 2, false>)
/* ... with logAdd merges */
#define FLTX_G36(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 5, 2, 1, 1, true>)
#define FLTX_G37(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 5, 4, 2, 2, true>)
#define FLTX_G38(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 10, 8, 2, 4, true>)
#define FLTX_G39(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 11, 2, 1, 1, true>)
#define FLTX_G40(W) FLTX_INST(fltx_decode_kernel_tmlane<960, 11, 4, 2, 2, true>)

#ifdef FLTX_INST_W
#define FLTX_CAT2_(a, b) a##b
#define FLTX_CAT_(a, b) FLTX_CAT2_(a, b)
FLTX_CAT_(FLTX_G, FLTX_INST_G)(FLTX_INST_W)
#undef FLTX_CAT_
#undef FLTX_CAT2_
#else
#define FLTX_ALLG(W) FLTX_G1(W) FLTX_G2(W) FLTX_G3(W) FLTX_G4(W) FLTX_G5(W) FLTX_G6(W) FLTX_G7(W) FLTX_G8(W) FLTX_G9(W)
FLTX_ALLG(64)
FLTX_ALLG(128)
FLTX_ALLG(256)
FLTX_ALLG(512)
FLTX_ALLG(1024)
FLTX_G10(0)
FLTX_G11(0)
FLTX_G12(0)
FLTX_G13(0)
FLTX_G14(0)
FLTX_G15(0)
FLTX_G16(0)
FLTX_G17(0)
FLTX_G18(0)
FLTX_G19(0)
FLTX_G20(0)
FLTX_G21(0)
FLTX_G22(0)
FLTX_G23(0)
FLTX_G24(0)
FLTX_G25(0)
FLTX_G26(0)
FLTX_G27(0)
FLTX_G28(0)
FLTX_G29(0)
FLTX_G30(0)
FLTX_G31(0)
FLTX_G32(0)
FLTX_G33(0)
FLTX_G34(0)
FLTX_G35(0)
FLTX_G36(0)
FLTX_G37(0)
FLTX_G38(0)
FLTX_G39(0)
FLTX_G40(0)
#undef FLTX_ALLG
#endif
#undef FLTX_G1
#undef FLTX_G2
#undef FLTX_G3
#undef FLTX_G4
#undef FLTX_G5
#undef FLTX_G6
#undef FLTX_G7
#undef FLTX_G8
#undef FLTX_G9
#undef FLTX_G10
#undef FLTX_G11
#undef FLTX_G12
#undef FLTX_G13
#undef FLTX_G14
#undef FLTX_G15
#undef FLTX_G16
#undef FLTX_G17
#undef FLTX_G18
#undef FLTX_G19
#undef FLTX_G20
#undef FLTX_G21
#undef FLTX_G22
#undef FLTX_G23
#undef FLTX_G24
#undef FLTX_G31
#undef FLTX_G32
#undef FLTX_G33
#undef FLTX_G34
#undef FLTX_G35
#undef FLTX_G36
#undef FLTX_G37
#undef FLTX_G38
#undef FLTX_G39
#undef FLTX_G40
#undef FLTX_G25
#undef FLTX_G26
#undef FLTX_G27
#undef FLTX_G28
#undef FLTX_G29
#undef FLTX_G30
#undef FLTX_TLANE_SET
#undef FLTX_MLANE_SET
#undef FLTX_YLANE_SET
#undef FLTX_XLANE_SET
#undef FLTX_XLANE_LA_SET
#undef FLTX_SLANE_SET
