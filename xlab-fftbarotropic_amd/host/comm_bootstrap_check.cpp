// comm_bootstrap_check.cpp -- command-line face of host/comm_bootstrap.hpp for the CPU suite (tests/test_host_cpp.py):
//   comm_bootstrap_check.out prepare PATH
//   comm_bootstrap_check.out publish PATH TOKEN BYTE          (an id of FB_UNIQUE_ID_BYTES bytes, all equal to BYTE)
//   comm_bootstrap_check.out await   PATH TOKEN TIMEOUT_S [MAX_AGE_S]   -> prints "id BYTE" or "timeout"
// No GPU call: what is under test is which record a rank > 0 accepts (never a leftover of another launch).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "comm_bootstrap.hpp"

int main(int argc, char **argv)
{
    const size_t ID = 128;
    if (argc >= 3 && !strcmp(argv[1], "prepare")) { fbcomm::prepare(argv[2]); return 0; }
    if (argc >= 5 && !strcmp(argv[1], "publish")) {
        char id[ID];
        memset(id, atoi(argv[4]), sizeof id);
        return fbcomm::publish(argv[2], argv[3], id, sizeof id) ? 0 : 1;
    }
    if (argc >= 5 && !strcmp(argv[1], "await")) {
        char id[ID];
        const long max_age = argc >= 6 ? atol(argv[5]) : 60;
        if (!fbcomm::await(argv[2], argv[3], id, sizeof id, atol(argv[4]), max_age)) { printf("timeout\n"); return 3; }
        for (size_t i = 1; i < ID; ++i) if (id[i] != id[0]) { printf("torn\n"); return 4; }
        printf("id %d\n", (int)(unsigned char)id[0]);
        return 0;
    }
    fprintf(stderr, "usage: prepare PATH | publish PATH TOKEN BYTE | await PATH TOKEN TIMEOUT_S [MAX_AGE_S]\n");
    return 2;
}
