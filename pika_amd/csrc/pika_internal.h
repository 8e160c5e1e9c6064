// pika_amd/csrc/pika_internal.h -- declarations shared between the translation units of libpika_amd.so (not part of
// the C ABI).
#ifndef PIKA_INTERNAL_H
#define PIKA_INTERNAL_H

// The device word registered by pika_set_dropout_salt (include/pika_gemm.h), or nullptr: every kernel that takes a
// dropout seed adds *salt to it, so a launch sequence replayed from a hipGraph draws new masks per replay.
const unsigned *pika_internal_dropout_salt();

#endif
