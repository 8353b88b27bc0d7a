// Forward-with-save instantiation of the fused network kernel (see the DHAUG_MLP_SAVE_TU section of dhaug_mlp.hip): the
// same source, a second translation unit with its own compiler flags.
#define DHAUG_MLP_SAVE_TU
#include "dhaug_mlp.hip"
