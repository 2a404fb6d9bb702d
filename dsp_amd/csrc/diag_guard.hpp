// diag_guard.hpp -- timing-only probes and diagnostic modes of the kernels (instruction-count probes, phase stamps, role subsets: the
// builds behind the records under profiles/) exist only in builds that say so: every such macro needs -DDSP_AMD_DIAG beside it
// (tools/mkvariant.sh NAME -DDSP_AMD_DIAG -D<probe>).  A product build cannot switch one on by accident.
#pragma once
#if !defined(DSP_AMD_DIAG)
#if defined(DSP_IIR_DIAG_NO_STORE) || (defined(DSP_PRE_DIAG) && DSP_PRE_DIAG != 0) || (defined(DSP_PAIR_DIAG) && DSP_PAIR_DIAG != 0) ||            \
    (defined(DSP_DIAG_MODE) && DSP_DIAG_MODE != 0) || defined(DSP_DIAG_NO_POOL) || defined(DSP_DIAG_NO_SVM) || defined(DSP_DIAG_NO_SVMTAIL) ||     \
    defined(DSP_DIAG_NO_POOLTILE) || defined(DSP_DIAG_NO_POOLFINISH) || defined(DSP_DIAG_SNOPS) || defined(DSP_DIAG_VNOPS) ||                      \
    defined(DSP_PF_STAMPS) || defined(DSP_RC_STAMPS) || defined(SC_DIAG) || (defined(SC_ROLES) && SC_ROLES != 7) || defined(SC_PRIO)
#error "timing-only probes / diagnostic modes need -DDSP_AMD_DIAG (dsp_amd/csrc/diag_guard.hpp)"
#endif
#endif
