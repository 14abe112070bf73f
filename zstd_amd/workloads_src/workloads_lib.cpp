// workloads_lib.cpp — bench / test INPUT GENERATORS (not product): built into zstd_amd/libzhip_workloads.so by zstd_amd/build.py.
// libzstd_hip.so exports nothing from here.
#include "zhip_datagen.h"

extern "C" void zhip_wl_datagen(void* buffer, size_t size, double matchProba, double litProba, unsigned seed, int streamMode)
{
    zhip::datagen(buffer, size, matchProba, litProba, seed, streamMode);
}

// ---------------------------------------------------------------- GitHub-user shaped JSON records (BASELINE configs[4] stand-in)
// The same record shape as zstd_amd/workloads.py:github_like_records (the generator the committed dictionary was trained on), written
// natively so that MILLIONS of distinct records can be made in seconds.  Own RNG: the records are not the Python generator's.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
namespace {
struct Rng { uint64_t s; uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }
             uint32_t below(uint32_t n) { return (uint32_t)((next() >> 33) % n); } double unit() { return (double)(next() >> 11) / 9007199254740992.0; } };
const char* const kSyll[] = {"an","ber","co","de","el","fi","go","ha","in","jo","ka","lu","mi","no","or","pa","qu","ri","so","ta","ul","vi","wa","xe","yo","zu",
                             "dev","code","hub","git","lab","sys","net","bit","io","x"};
const char* const kPlaces[] = {"San Francisco, CA","Berlin, Germany","London","Tokyo, Japan","Paris","Bangalore, India","Sao Paulo","Toronto, Canada","Seattle, WA",nullptr};
const char* const kComps[] = {"@github","Google","Microsoft","Red Hat",nullptr,nullptr,"@facebook","ACME Corp","University of Somewhere",nullptr};
const char* const kBios[] = {"Software engineer.","I build things for the web.",nullptr,nullptr,"Open source enthusiast","Student","Full-stack developer and coffee drinker",nullptr};
inline char* put(char* p, const char* s) { size_t const n = strlen(s); memcpy(p, s, n); return p + n; }
inline char* putq(char* p, const char* s) { if (!s) return put(p, "null"); *p++ = '"'; p = put(p, s); *p++ = '"'; return p; }
inline char* putu(char* p, unsigned long long v, int width = 0) { char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n < width) t[n++] = '0'; while (n) *p++ = t[--n]; return p; }
}
// writes nRecords records back to back into flat (capacity cap bytes) and their offsets (nRecords + 1) into offs; returns the bytes used, 0 on overflow
extern "C" size_t zhip_wl_github_records(char* flat, size_t cap, unsigned long long* offs, size_t nRecords, unsigned seed)
{
    Rng r{0x9E3779B97F4A7C15ULL ^ ((uint64_t)seed * 0xD1B54A32D192ED03ULL + 4242)};
    for (int i = 0; i < 8; i++) r.next();
    char* p = flat;
    for (size_t k = 0; k < nRecords; k++) {
        if ((size_t)(p - flat) + 2048 > cap) return 0;
        offs[k] = (unsigned long long)(p - flat);
        char login[64]; { char* q = login; uint32_t const ns = 2 + r.below(3); for (uint32_t i = 0; i < ns; i++) q = put(q, kSyll[r.below(36)]);
                          if (r.unit() < 0.4) { q = putu(q, r.below(999)); }
                          *q = 0; }
        unsigned long long const uid = 1 + r.next() % 89999999ULL;
        char name[96]; bool const hasName = r.unit() >= 0.3;
        if (hasName) { char* q = name; q = put(q, login); name[0] = (char)(name[0] >= 'a' && name[0] <= 'z' ? name[0] - 32 : name[0]); *q++ = ' ';
                       const char* s = kSyll[r.below(26)]; char* q0 = q; q = put(q, s); *q0 = (char)(*q0 - 32); q = put(q, "son"); *q = 0; }
        p = put(p, "{\"login\":\""); p = put(p, login); p = put(p, "\",\"id\":"); p = putu(p, uid); p = put(p, ",\"node_id\":\"MDQ6VXNlcj"); p = putu(p, uid);
        p = put(p, "\",\"avatar_url\":\"https://avatars.githubusercontent.com/u/"); p = putu(p, uid); p = put(p, "?v=4\",\"gravatar_id\":\"\",");
        p = put(p, "\"url\":\"https://api.github.com/users/"); p = put(p, login); p = put(p, "\",\"html_url\":\"https://github.com/"); p = put(p, login);
        static const char* const tails[] = {"followers_url\":\"https://api.github.com/users/%/followers", "following_url\":\"https://api.github.com/users/%/following{/other_user}",
            "gists_url\":\"https://api.github.com/users/%/gists{/gist_id}", "starred_url\":\"https://api.github.com/users/%/starred{/owner}{/repo}",
            "subscriptions_url\":\"https://api.github.com/users/%/subscriptions", "organizations_url\":\"https://api.github.com/users/%/orgs",
            "repos_url\":\"https://api.github.com/users/%/repos", "events_url\":\"https://api.github.com/users/%/events{/privacy}",
            "received_events_url\":\"https://api.github.com/users/%/received_events"};
        for (const char* t : tails) { p = put(p, "\",\""); for (const char* c = t; *c; c++) { if (*c == '%') p = put(p, login); else *p++ = *c; } }
        p = put(p, "\",\"type\":\"User\",\"site_admin\":false,\"name\":"); p = putq(p, hasName ? name : nullptr);
        p = put(p, ",\"company\":"); p = putq(p, kComps[r.below(10)]);
        p = put(p, ",\"blog\":\""); if (r.unit() < 0.2) { p = put(p, "https://"); p = put(p, login); p = put(p, ".dev"); }
        p = put(p, "\",\"location\":"); p = putq(p, kPlaces[r.below(10)]);
        p = put(p, ",\"email\":null,\"hireable\":"); p = put(p, r.unit() < 0.1 ? "true" : "null");
        p = put(p, ",\"bio\":"); p = putq(p, kBios[r.below(8)]);
        p = put(p, ",\"twitter_username\":null,\"public_repos\":"); p = putu(p, r.below(300)); p = put(p, ",\"public_gists\":"); p = putu(p, r.below(40));
        p = put(p, ",\"followers\":"); p = putu(p, r.below(5000)); p = put(p, ",\"following\":"); p = putu(p, r.below(500));
        p = put(p, ",\"created_at\":\""); p = putu(p, 2008 + r.below(16), 4); *p++ = '-'; p = putu(p, 1 + r.below(12), 2); *p++ = '-'; p = putu(p, 1 + r.below(28), 2);
        *p++ = 'T'; p = putu(p, r.below(24), 2); *p++ = ':'; p = putu(p, r.below(60), 2); *p++ = ':'; p = putu(p, r.below(60), 2);
        p = put(p, "Z\",\"updated_at\":\"2024-"); p = putu(p, 1 + r.below(12), 2); *p++ = '-'; p = putu(p, 1 + r.below(28), 2);
        *p++ = 'T'; p = putu(p, r.below(24), 2); *p++ = ':'; p = putu(p, r.below(60), 2); *p++ = ':'; p = putu(p, r.below(60), 2);
        p = put(p, "Z\"}");
    }
    offs[nRecords] = (unsigned long long)(p - flat);
    return (size_t)(p - flat);
}
