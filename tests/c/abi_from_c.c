/* The C ABI seen from plain C99: the header compiles with -pedantic, the library links, and without a GPU the first
 * compute-side call fails with an error code and a message instead of aborting (tests/test_host_cpu.py builds and
 * runs this file).  With a GPU it creates and destroys an engine. */
#include "remora_hip.h"
#include <stdio.h>
#include <string.h>

int main(void) {
    rmr_engine *e = NULL;
    rmr_model_desc d;
    int rc;
    printf("version=%s\n", rmr_version());
    memset(&d, 0, sizeof d);
    d.arch = RMR_ARCH_CONV_LSTM;
    d.size = 64;
    d.kmer_len = 9;
    d.num_out = 2;
    d.chunk_len = 100;
    printf("weights=%lu\n", (unsigned long)rmr_model_weight_count(&d));
    rc = rmr_engine_create(0, NULL, 0, &e);
    if (rc != 0) {
        printf("engine_create rc=%d error=%s\n", rc, rmr_last_error());
        return e == NULL ? 0 : 1;
    }
    rc = rmr_engine_synchronize(e);
    rmr_engine_destroy(e);
    printf("engine ok rc=%d\n", rc);
    return rc;
}
