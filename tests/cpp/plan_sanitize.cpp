// tests/cpp/plan_sanitize.cpp -- the launch-plan builder (csrc/mapn_sym_plan.cpp: host-only code) under AddressSanitizer +
// UndefinedBehaviorSanitizer: a sweep of shapes through mapn_sym_plan_describe -- block counts, parts, tapers, wave biases, the three
// XCD modes with skewed weights, sharded launches -- with output arrays of EXACTLY the size the first call reports, so that a write
// past `windows_capacity` / `tables_capacity` is a heap overflow the sanitizer sees.  (GPU sanitizers do not exist on this pool; this
// is the part of the product that is plain C++.)  Built and run by tests/test_sym_cpu.py.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mapn_tuning.h"

int main()
{
    unsigned ok = 0, refused = 0;
    const uint32_t nbs[] = {1, 2, 3, 7, 8, 16, 27, 63, 64, 65, 72, 128, 256, 1024};
    const uint32_t weights[3][8] = {{1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024}, {998, 1021, 986, 1024, 995, 1016, 979, 1016}, {1, 4096, 17, 4096, 900, 1024, 2048, 3}};
    for (uint32_t nb : nbs)
        for (uint32_t parts : {1u, 2u, 4u, 7u, 8u, 16u, 32u})
            for (uint32_t taper : {0u, 1u, 2u})
                for (uint32_t waves : {4u, 8u})
                    for (uint32_t bias : {0u, 1u, 2u})
                        for (uint32_t mode = 0; mode < 3; mode++)
                            for (uint32_t shard : {0u, 2u, 8u}) {
                                if (shard && nb % shard) continue;
                                const uint32_t blocks = shard ? nb / shard : 0u, a0 = shard ? blocks * (shard - 1u) : 0u;
                                const uint32_t hi = bias == 0 ? 1u : bias == 1 ? 3u : 10u, lo = bias == 0 ? 1u : bias == 1 ? 1u : 3u;
                                const uint32_t t1 = taper ? parts / 4u : 0u, t2 = taper == 2 ? parts / 4u : 0u;
                                const uint32_t groups = nb > 64 ? 8u : 0u;              // windows of partner distance for the larger jobs
                                mapn_sym_plan_info info;
                                memset(&info, 0, sizeof info);
                                const uint32_t *w = mode ? weights[1 + (nb + parts) % 2] : nullptr;
                                int rc = mapn_sym_plan_describe(nb, groups, parts, t1, t2, waves, hi, lo, w, blocks, a0, mode == 1 ? 1u : 0u, &info, nullptr, 0, nullptr, 0);
                                if (rc != 0) { refused++; continue; }
                                const size_t nwin = info.windows, ntab = (size_t)info.windows * info.table_stride + info.wgmap_entries;
                                std::vector<uint32_t> win(4 * nwin), tab(ntab);          // exactly as large as reported
                                rc = mapn_sym_plan_describe(nb, groups, parts, t1, t2, waves, hi, lo, w, blocks, a0, mode == 1 ? 1u : 0u, &info, win.data(), 4u * (uint32_t)nwin, tab.data(),
                                                            (uint32_t)ntab);
                                if (rc != 0) { printf("second call failed: nb %u parts %u: %s\n", nb, parts, info.error); return 1; }
                                if (ntab > 1) {                                           // one element short: refused, nothing written past the end
                                    std::vector<uint32_t> small(ntab - 1);
                                    rc = mapn_sym_plan_describe(nb, groups, parts, t1, t2, waves, hi, lo, w, blocks, a0, mode == 1 ? 1u : 0u, &info, win.data(), 4u * (uint32_t)nwin, small.data(),
                                                                (uint32_t)(ntab - 1));
                                    if (rc == 0) { printf("a table array one word short was accepted: nb %u parts %u\n", nb, parts); return 1; }
                                }
                                ok++;
                            }
    printf("plans built %u, shapes refused %u\n", ok, refused);
    return ok > 500 ? 0 : 1;
}
