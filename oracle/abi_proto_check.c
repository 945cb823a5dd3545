/* abi_proto_check.c -- TEST INFRASTRUCTURE: the reference's own headers (init.h, tools.h, channel.h, bubble_decoder.h, where they
 * lie under $(REF)/include) and include/nbldpc_ref_api.h in ONE translation unit, our header declaring its prototypes on the
 * reference's types (NBLDPC_REF_HAVE_STRUCT_H).  Every symbol both sides declare is then declared twice, and C requires the two
 * declarations to be compatible: a signature of ours that drifted from the reference's is a compile error here. */
#include "struct.h"
#include "init.h"
#include "tools.h"
#include "channel.h"
#include "bubble_decoder.h"

#define NBLDPC_REF_HAVE_STRUCT_H
#include "nbldpc_ref_api.h"

int abi_proto_check_ok = 1;
