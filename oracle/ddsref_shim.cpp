// TEST INFRASTRUCTURE: extern "C" entry points over the REFERENCE's PVM codec
// (compiled from /root/reference/src/ddsbase.cpp where it lies; see Makefile).
// Used only by tests/golden/make_pvm_fixtures.py to mint .pvm fixtures and by the
// PVM parity tests when oracle/_ref/libddsref.so is present.
#include <stdlib.h>
#include <string.h>

#include "ddsbase.h"   // from the reference include dir (-I)

extern "C" {

void ddsref_write_pvm(const char *fn, const unsigned char *vol, unsigned w, unsigned h, unsigned d,
                      unsigned comps, float sx, float sy, float sz, const char *desc)
{
    // writePVMvolume frees nothing it did not allocate but DDS-encodes in place: pass a copy
    size_t n = (size_t)w * h * d * comps;
    unsigned char *copy = (unsigned char *)malloc(n);
    memcpy(copy, vol, n);
    writePVMvolume(fn, copy, w, h, d, comps, sx, sy, sz, (unsigned char *)desc);
    free(copy);
}

// returns malloc'ed payload (caller frees with ddsref_free) or NULL
unsigned char *ddsref_read_pvm(const char *fn, unsigned *w, unsigned *h, unsigned *d, unsigned *comps,
                               float *sx, float *sy, float *sz)
{
    return readPVMvolume(fn, w, h, d, comps, sx, sy, sz);
}

unsigned ddsref_checksum(const unsigned char *data, unsigned bytes) { return checksum((unsigned char *)data, bytes); }

void ddsref_free(void *p) { free(p); }
}
