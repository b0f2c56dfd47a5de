// comm_bootstrap.hpp -- hand-over of the RCCL unique id between the ranks of barotropic_main.out through a shared file.
// No reference counterpart (the reference is single-process); host logic only, no GPU call, so that the CPU suite can
// cover it (host/comm_bootstrap_check.cpp, tests/test_host_cpp.py).
//
// File = 8 bytes magic + 64 bytes launch token (zero padded) + the id.  Rank 0 removes whatever lies at the path before
// it does anything else and publishes with write-to-temporary + rename, so a reader never sees a partial record.  A
// rank > 0 takes a record only if its token equals its own --launch-token (the same string on every rank of ONE launch:
// a job id, a start time); a leftover of another launch -- which would hand it a dead ncclUniqueId and leave it hanging in
// ncclCommInitRank -- is skipped.  Without a token the record must not be older than `max_age_s` seconds at the moment
// the reader started (start the ranks within that window, or pass a token).
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>

namespace fbcomm {
constexpr size_t TOKEN_BYTES = 64;
constexpr char MAGIC[8] = {'F', 'B', 'C', 'O', 'M', 'M', '1', '\0'};

inline void prepare(const std::string &path) { unlink(path.c_str()); unlink((path + ".tmp").c_str()); }   // rank 0, first thing

inline bool publish(const std::string &path, const std::string &token, const char *id, size_t id_bytes)
{
    if (token.size() >= TOKEN_BYTES) return false;
    char tok[TOKEN_BYTES];
    memset(tok, 0, sizeof tok);
    memcpy(tok, token.data(), token.size());
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(MAGIC, 1, sizeof MAGIC, f) == sizeof MAGIC && fwrite(tok, 1, sizeof tok, f) == sizeof tok &&
                    fwrite(id, 1, id_bytes, f) == id_bytes;
    if (fclose(f) != 0 || !ok) { unlink(tmp.c_str()); return false; }
    return rename(tmp.c_str(), path.c_str()) == 0;
}

// one look at the file: 1 = id taken, 0 = nothing acceptable there (absent, partial, other launch's token, too old)
inline int try_read(const std::string &path, const std::string &token, char *id, size_t id_bytes, time_t reader_start, long max_age_s)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    char magic[sizeof MAGIC], tok[TOKEN_BYTES];
    struct stat st;
    const bool ok = fread(magic, 1, sizeof magic, f) == sizeof magic && fread(tok, 1, sizeof tok, f) == sizeof tok &&
                    fread(id, 1, id_bytes, f) == id_bytes && fstat(fileno(f), &st) == 0;
    fclose(f);
    if (!ok || memcmp(magic, MAGIC, sizeof MAGIC) != 0) return 0;
    char want[TOKEN_BYTES];
    memset(want, 0, sizeof want);
    memcpy(want, token.data(), token.size() < TOKEN_BYTES ? token.size() : TOKEN_BYTES - 1);
    if (memcmp(tok, want, TOKEN_BYTES) != 0) return 0;                       // another launch's record
    if (token.empty() && st.st_mtime + max_age_s < reader_start) return 0;   // no token: only a fresh record counts
    return 1;
}

// ranks > 0: poll until rank 0's record of THIS launch appears; false after timeout_s seconds
inline bool await(const std::string &path, const std::string &token, char *id, size_t id_bytes, long timeout_s, long max_age_s = 60)
{
    const time_t start = time(nullptr);
    for (;;) {
        if (try_read(path, token, id, id_bytes, start, max_age_s)) return true;
        if (time(nullptr) - start > timeout_s) return false;
        usleep(50000);
    }
}
}  // namespace fbcomm
