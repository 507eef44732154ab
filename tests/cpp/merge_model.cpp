// merge_model.cpp -- the two forms of the Huffman builder's two-queue merge held against each other on the CPU: serial() is the reference's
// loop (fpng.cpp:645-651) as build_dynamic_kernel ran it on one lane until round 5, bulk() what the wave does now (kernels.hip, dev_build_table: runs
// of picks whose choices do not depend on what the run makes are taken at once, 64 lanes).  Same parents and weights for every sorted key set --
// shapes of real histograms, and random ones whose sums wrap at 16 bits.  Prints the steps both take.   usage: merge_model [burst] [key files ...]
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#include <cstring>
struct Out { std::vector<int> lparent, iparent; std::vector<uint32_t> iw; };
static Out serial(const std::vector<uint32_t> &skey)
{
    const uint32_t used = skey.size();
    Out o; o.lparent.assign(used, -1); o.iparent.assign(used, -1); o.iw.assign(used, 0);
    uint32_t leaf = 0, root = 0, made = 0;
    while (made + 1 < used) {
        uint32_t wsum = 0;
        for (int k = 0; k < 2; k++) {
            if (leaf >= used || (root < made && o.iw[root] < skey[leaf])) { wsum += o.iw[root]; o.iparent[root++] = (int)made; }
            else { wsum += skey[leaf]; o.lparent[leaf++] = (int)made; }
        }
        o.iw[made++] = wsum & 0xFFFFu;
    }
    return o;
}
static long g_bulk_it, g_mixed, g_serial_steps;
static Out bulk(const std::vector<uint32_t> &skey, int burst)
{
    const uint32_t used = skey.size();
    Out o; o.lparent.assign(used, -1); o.iparent.assign(used, -1); o.iw.assign(used, 0);
    uint32_t leaf = 0, root = 0, made = 0;
    while (made + 1 < used) {
        const bool q = root < made, have_leaf = leaf < used;
        const uint32_t H = q ? o.iw[root] : 0, K = have_leaf ? skey[leaf] : 0;
        const bool first_internal = !have_leaf || (q && H < K);
        uint32_t run = 0;
        if (!first_internal) {
            for (uint32_t i = 0; i < 64; i++) { const uint32_t li = leaf + i; if (li < used && !(q && H < skey[li])) run++; else break; }
            if (!q && run > 2) run = 2;
            const uint32_t pairs = run >> 1;
            if (pairs) {
                for (uint32_t l = 0; l < pairs; l++) { const uint32_t node = made + l; o.iw[node] = (skey[leaf + 2 * l] + skey[leaf + 2 * l + 1]) & 0xFFFFu; o.lparent[leaf + 2 * l] = o.lparent[leaf + 2 * l + 1] = (int)node; }
                leaf += 2 * pairs; made += pairs; g_bulk_it++; continue;
            }
        } else {
            for (uint32_t i = 0; i < 64; i++) { const uint32_t ri = root + i; if (ri < made && (!have_leaf || o.iw[ri] < K)) run++; else break; }
            const uint32_t pairs = run >> 1;
            if (pairs) {
                for (uint32_t l = 0; l < pairs; l++) { const uint32_t node = made + l; o.iw[node] = (o.iw[root + 2 * l] + o.iw[root + 2 * l + 1]) & 0xFFFFu; o.iparent[root + 2 * l] = o.iparent[root + 2 * l + 1] = (int)node; }
                root += 2 * pairs; made += pairs; g_bulk_it++; continue;
            }
        }
        g_mixed++;
        for (int s = 0; s < burst && made + 1 < used; s++) {
            uint32_t wsum = 0;
            for (int k = 0; k < 2; k++) {
                if (leaf >= used || (root < made && o.iw[root] < skey[leaf])) { wsum += o.iw[root]; o.iparent[root++] = (int)made; }
                else { wsum += skey[leaf]; o.lparent[leaf++] = (int)made; }
            }
            o.iw[made++] = wsum & 0xFFFFu; g_serial_steps++;
        }
    }
    return o;
}
// The length limiter behind the merge (reference fpng.cpp:663-674): serial as the reference, and with the iterations that take their code
// from level max_len - 1 applied in one go (the kernel's form).  In: the leaves' depths from a merge.
static long g_kraft_serial_it, g_kraft_batched_it;
static std::vector<int> depth_counts(const Out &o, uint32_t used)
{
    std::vector<int> nc(40, 0);
    for (uint32_t i = 0; i < used; i++) {
        int node = o.lparent[i], d = 1;
        while (node != (int)used - 2) node = o.iparent[node], d++;
        nc[d > 39 ? 39 : d]++;
    }
    return nc;
}
static bool kraft_both(std::vector<int> nc, int max_len)
{
    for (int i = max_len + 1; i < 40; i++) nc[max_len] += nc[i], nc[i] = 0;
    uint32_t total = 0;
    for (int i = max_len; i > 0; i--) total += (uint32_t)nc[i] << (max_len - i);
    std::vector<int> a = nc, b = nc;
    uint32_t ta = total, tb = total;
    while (ta != (1u << max_len)) {
        a[max_len]--;
        for (int i = max_len - 1; i > 0; i--)
            if (a[i]) { a[i]--; a[i + 1] += 2; break; }
        ta--; g_kraft_serial_it++;
    }
    // the kernel's form: the reference walks depth-first -- a code taken from level l is split down to max_len (2^(max_len - l) - 1 iterations,
    // after which the levels between are empty again and level max_len has gained one code) before the next code of level l is touched: as many
    // whole walks as the excess pays for are applied in one go; what is left over takes single iterations (at most one per level)
    while (tb != (1u << max_len)) {
        int l = 0;
        for (int i = max_len - 1; i > 0; i--) if (b[i]) { l = i; break; }
        const uint32_t excess = tb - (1u << max_len);
        if (l) {
            const uint32_t cost = (1u << (max_len - l)) - 1u;
            const uint32_t k = std::min<uint32_t>((uint32_t)b[l], excess / cost);
            if (k) { b[l] -= (int)k; b[max_len] += (int)k; tb -= k * cost; g_kraft_batched_it++; continue; }
            b[max_len]--; b[l]--; b[l + 1] += 2; tb--;
        } else {
            b[max_len]--; tb--;
        }
        g_kraft_batched_it++;
    }
    return a == b;
}
int main(int argc, char **argv)
{
    const int burst = argc > 1 ? atoi(argv[1]) : 1;
    std::mt19937 rng(12345);
    long trials = 0;
    auto check = [&](std::vector<uint32_t> k, const char *what, bool report) {
        std::sort(k.begin(), k.end());
        g_bulk_it = g_mixed = g_serial_steps = 0;
        Out a = serial(k), b = bulk(k, burst);
        if (a.lparent != b.lparent || a.iparent != b.iparent || a.iw != b.iw) { printf("MISMATCH %s n=%zu\n", what, k.size()); exit(1); }
        if (k.size() >= 2) {
            g_kraft_serial_it = g_kraft_batched_it = 0;
            const std::vector<int> nc = depth_counts(a, (uint32_t)k.size());
            // (the kernel builds the 19-symbol table with max_len 7 and the 288-symbol one with 12; both limits on every tree with enough leaves)
            if (k.size() <= 128 && !kraft_both(nc, 7)) { printf("KRAFT MISMATCH (7) %s n=%zu\n", what, k.size()); exit(1); }
            if (!kraft_both(nc, 12)) { printf("KRAFT MISMATCH (12) %s n=%zu\n", what, k.size()); exit(1); }
            if (report) printf("   length limiter: %ld iterations serial, %ld with whole depth-first walks in one go\n", g_kraft_serial_it, g_kraft_batched_it);
        }
        trials++;
        if (report) printf("%-28s n=%3zu: serial steps %3zu (%6ld cycles at 470) | bulk iterations %3ld, fall-throughs %3ld, serial steps %3ld -> ~%6ld cycles (300 / 250 / 470)\n", what, k.size(), k.size() - 1,
                           (long)(k.size() - 1) * 470, g_bulk_it, g_mixed, g_serial_steps, g_bulk_it * 300 + g_mixed * 250 + g_serial_steps * 470);
    };
    // shapes
    { std::vector<uint32_t> k(260, 1); for (int i = 0; i < 24; i++) k.push_back(3 + i * i * 37); check(k, "grad-like (260 ones + 24)", true); }
    { std::vector<uint32_t> k; for (int i = 0; i < 288; i++) k.push_back(220 + rng() % 20); check(k, "uniform", true); }
    { std::vector<uint32_t> k; double f = 1; for (int i = 0; i < 140; i++) { k.push_back((uint32_t)f), k.push_back((uint32_t)f); f *= 1.07; if (f > 20000) f = 20000; } check(k, "laplacian ratio 1.07", true); }
    { std::vector<uint32_t> k; double f = 1; for (int i = 0; i < 100; i++) { k.push_back((uint32_t)f), k.push_back((uint32_t)f); f *= 1.2; if (f > 30000) f = 30000; } check(k, "laplacian ratio 1.2", true); }
    { std::vector<uint32_t> k; uint32_t a = 1, b = 1; for (int i = 0; i < 22; i++) { k.push_back(a); uint32_t c = a + b; a = b; b = c; } check(k, "fibonacci (22)", true); }
    { std::vector<uint32_t> k; double f = 1; for (int i = 0; i < 288; i++) { k.push_back((uint32_t)f); f *= 1.035; } check(k, "geometric 1.035 (288)", true); }
    { std::vector<uint32_t> k; for (int i = 0; i < 288; i++) k.push_back(1 + i); check(k, "linear 1..288", true); }
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "r"); std::vector<uint32_t> k; unsigned v; while (f && fscanf(f, "%u", &v) == 1) k.push_back(v); if (f) fclose(f);
        check(k, argv[a], true);
    }
    // exhaustive: every multiset of up to 9 keys from a small set (ties everywhere, sums that wrap at 16 bits)
    {
        const uint32_t vals[] = {1, 2, 3, 5, 30000, 40000, 65535};
        const int nv = sizeof vals / sizeof vals[0];
        for (int n = 2; n <= 9; n++) {
            std::vector<int> idx(n, 0);
            for (;;) {
                std::vector<uint32_t> k(n);
                for (int i = 0; i < n; i++) k[i] = vals[idx[i]];
                check(k, "exhaustive", false);
                int p = n - 1; // next non-decreasing index vector
                while (p >= 0 && idx[p] == nv - 1) p--;
                if (p < 0) break;
                const int v = idx[p] + 1;
                for (int i = p; i < n; i++) idx[i] = v;
            }
        }
    }
    // random: sizes, value ranges (incl. keys whose sums wrap at 16 bits)
    for (int t = 0; t < 150000; t++) {
        const uint32_t n = 2 + rng() % 287;
        const int mode = rng() % 6;
        std::vector<uint32_t> k(n);
        for (auto &v : k) {
            switch (mode) {
            case 0: v = 1 + rng() % 3; break;
            case 1: v = 1 + rng() % 65535; break;
            case 2: v = 1 + (rng() % 16) * (rng() % 16) * (rng() % 256); break;
            case 3: v = 1u << (rng() % 16); break;
            case 4: v = 1 + rng() % 300; break;
            default: v = (rng() % 4) ? 1 : 1 + rng() % 65535; break;
            }
            if (v > 65535) v = 65535;
        }
        check(k, "random", false);
    }
    printf("%ld trials: bulk == serial (burst %d)\n", trials, burst);
    return 0;
}
