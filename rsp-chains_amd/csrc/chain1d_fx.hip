// The FIXED16 half of chain1d.hip (see "compiled twice" there): the same source with RSP_PART_FX, a second object.
#define RSP_PART_FX 1
#include "chain1d.hip"
