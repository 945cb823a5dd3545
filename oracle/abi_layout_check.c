/* abi_layout_check.c -- TEST INFRASTRUCTURE: compile-time proof that the struct types of include/nbldpc_ref_api.h (what
 * libems_hip.so is compiled against) have the layout of the reference's own include/struct.h:48-108 as a C compiler sees it.
 * Compiled by `make -C oracle abi_check` with -I$(REF)/include (the reference's header, where it lies) and -I../include
 * (ours, its three types renamed by macros).  It contains nothing but _Static_asserts: compiling IS the test. */
#include <stddef.h>

#include "struct.h"                      /* the reference's: code_t, table_t, decoder_t, softdata_t */
typedef code_t ref_code_t;
typedef table_t ref_table_t;
typedef decoder_t ref_decoder_t;
typedef softdata_t ref_softdata_t;

#define code_t ems_code_t
#define table_t ems_table_t
#define decoder_t ems_decoder_t
#define softdata_t ems_softdata_t
#include "nbldpc_ref_api.h"             /* ours */
#undef code_t
#undef table_t
#undef decoder_t
#undef softdata_t

#define SAME_SIZE(a, b) _Static_assert(sizeof(a) == sizeof(b), "sizeof(" #a ") != sizeof(" #b ")")
#define SAME_FIELD(a, b, f)                                                                              \
    _Static_assert(offsetof(a, f) == offsetof(b, f), "offset of " #f " differs between " #a " and " #b); \
    _Static_assert(sizeof(((a *)0)->f) == sizeof(((b *)0)->f), "size of " #f " differs between " #a " and " #b); \
    _Static_assert(__builtin_types_compatible_p(__typeof__(((a *)0)->f), __typeof__(((b *)0)->f)), "type of " #f " differs between " #a " and " #b)

SAME_SIZE(ref_softdata_t, ems_softdata_t);
_Static_assert(__builtin_types_compatible_p(ref_softdata_t, ems_softdata_t), "softdata_t");

SAME_SIZE(ref_code_t, ems_code_t);
SAME_FIELD(ref_code_t, ems_code_t, N);
SAME_FIELD(ref_code_t, ems_code_t, M);
SAME_FIELD(ref_code_t, ems_code_t, K);
SAME_FIELD(ref_code_t, ems_code_t, GF);
SAME_FIELD(ref_code_t, ems_code_t, logGF);
SAME_FIELD(ref_code_t, ems_code_t, mat);
SAME_FIELD(ref_code_t, ems_code_t, matValue);
SAME_FIELD(ref_code_t, ems_code_t, rowDegree);
SAME_FIELD(ref_code_t, ems_code_t, columnDegree);
SAME_FIELD(ref_code_t, ems_code_t, nbBranch);
SAME_FIELD(ref_code_t, ems_code_t, rate);
SAME_FIELD(ref_code_t, ems_code_t, matUT);
SAME_FIELD(ref_code_t, ems_code_t, Perm);

SAME_SIZE(ref_table_t, ems_table_t);
SAME_FIELD(ref_table_t, ems_table_t, BINGF);
SAME_FIELD(ref_table_t, ems_table_t, ADDGF);
SAME_FIELD(ref_table_t, ems_table_t, MULGF);
SAME_FIELD(ref_table_t, ems_table_t, DIVGF);

SAME_SIZE(ref_decoder_t, ems_decoder_t);
SAME_FIELD(ref_decoder_t, ems_decoder_t, N);
SAME_FIELD(ref_decoder_t, ems_decoder_t, nbMax);
SAME_FIELD(ref_decoder_t, ems_decoder_t, nbBranch);
SAME_FIELD(ref_decoder_t, ems_decoder_t, CtoV);
SAME_FIELD(ref_decoder_t, ems_decoder_t, intrinsic_LLR);
SAME_FIELD(ref_decoder_t, ems_decoder_t, intrinsic_GF);
SAME_FIELD(ref_decoder_t, ems_decoder_t, APP);
SAME_FIELD(ref_decoder_t, ems_decoder_t, M_VtoC_LLR);
SAME_FIELD(ref_decoder_t, ems_decoder_t, M_VtoC_GF);
SAME_FIELD(ref_decoder_t, ems_decoder_t, M_CtoV_LLR);
SAME_FIELD(ref_decoder_t, ems_decoder_t, M_CtoV_GF);

int abi_layout_check_ok = 1;
