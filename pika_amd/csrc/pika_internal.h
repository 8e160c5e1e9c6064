// pika_amd/csrc/pika_internal.h -- declarations shared between the translation units of libpika_amd.so (not part of
// the C ABI).
#ifndef PIKA_INTERNAL_H
#define PIKA_INTERNAL_H

// The device word registered by pika_set_dropout_salt (include/pika_gemm.h), or nullptr: every kernel that takes a
// dropout seed adds *salt to it, so a launch sequence replayed from a hipGraph draws new masks per replay.
const unsigned *pika_internal_dropout_salt();

// Tuning knobs of A/B runs (tools/pp_bench.py, tools/dw_bench.py, tools/dfc2_bench.py ...): an environment variable is
// read ONLY in a tuning build -- PIKA_HIPCC_EXTRA=-DPIKA_TUNING_KNOBS python -m pika_amd.build --force -- the shipped
// library reads no PIKA_* variable at all: every knob has the value its sweep chose (the default next to each use).
#ifdef PIKA_TUNING_KNOBS
#include <stdlib.h>
static inline const char *pika_knob(const char *name) { return getenv(name); }
#else
static inline const char *pika_knob(const char *) { return nullptr; }
#endif

#endif
