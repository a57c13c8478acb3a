// plan.h — host-side work partitioning for the per-frame kernels (pure C++, no HIP).
//
// The reference walks `bond type -> molecule instance` (topology/bond.rs:396-446) once per frame.
// On the GPU the same samples are re-grouped by WHERE THEIR ATOMS LIVE IN THE FRAME: a *tile* is a
// set of up to `block` samples whose atoms all fall into one contiguous window of the AoS frame
// [atom0, atom0 + n_window).  One workgroup owns one tile for a whole range of frames: it streams
// the window of every frame through LDS with 16-byte coalesced loads (each coordinate byte is
// fetched from HBM exactly once) and each thread keeps the running i64 sum of ITS sample in
// registers, so no cross-lane reduction or atomic is needed inside the frame loop.
// Samples whose own atoms span more than `max_window` atoms go to a direct-gather list.
#pragma once
#include <algorithm>
#include <map>
#include <tuple>
#include <cstdint>
#include <utility>
#include <vector>

#include "../../include/gorder_hip.h"

namespace gorder {

#ifndef GORDER_BLOCK
#define GORDER_BLOCK 256
#endif
constexpr uint32_t kBlock = GORDER_BLOCK;   // threads per workgroup = samples per tile (64 or 256)
constexpr uint32_t kMaxWindow = 1024;   // atoms per LDS window (12 KiB per staged frame)

struct Item {            // one AA/CG bond sample of a tile (12 bytes)
    uint16_t li, lj;     // atom indices relative to the tile's atom0 (p1, p2 of bond.rs:407-418)
    uint16_t lslot;      // index into the tile's slot list
    uint16_t pad;
    uint32_t mol;        // global molecule id (molecule type major) -> leaflet flag lookup
};
struct Tile {
    uint32_t atom0, n_window;
    uint32_t item0, n_items;
    uint32_t slot0, n_slots;   // range in Plan::tile_slots
};
struct DirectItem {
    uint32_t i, j, slot, mol;
};
struct UaItem {          // one united-atom carbon instance (uaorder.rs:911-915, 1050-1055), 16 bytes
    uint16_t l[4];       // helper1,target,helper2,- (CH3/CH2/CH1 unsat) or h1,h2,h3,target (CH1 sat), window-relative
    uint16_t lslot0;     // first local slot; hydrogen k uses lslot0 + k
    uint16_t kind;       // gorder_ua_kind_t
    uint32_t mol;
};

struct MapRun {          // `n` consecutive threads of a united-atom tile (from tid0) are the molecules of ONE slot:
    uint32_t tile, tid0, n, k;   // their hydrogen k.  Lets k_map_accumulate read a slot's staged samples in runs.
};

struct Plan {
    uint32_t n_atoms = 0, n_acc = 0, n_mol_total = 0;
    uint32_t max_window = 0;
    std::vector<Tile> tiles;
    std::vector<Item> items;
    std::vector<uint32_t> tile_slots;
    std::vector<DirectItem> direct;
    // united-atom carbons: same tiling idea, separate item type; no direct fallback (the 3-4 atoms of one
    // carbon are bonded neighbours — a tuple wider than 16-bit offsets reach is rejected)
    std::vector<Tile> ua_tiles;
    std::vector<UaItem> ua_items;
    std::vector<uint32_t> ua_tile_slots;
    std::vector<MapRun> ua_runs;            // grouped by accumulator slot:
    std::vector<uint32_t> ua_run_begin;     // [n_acc + 1] (CSR)
    // the same for bond tiles: `items_by_slot` = every tile's items re-ordered so that the molecules of one slot
    // sit on consecutive lanes (used by the scatter kernel only; K1 keeps the atom order of `items`)
    std::vector<Item> items_by_slot;
    std::vector<uint32_t> item_run, ua_item_run;   // per item of items_by_slot / ua_items: (tid0 << 16) | n of its run
    std::vector<MapRun> runs;
    std::vector<uint32_t> run_begin;
    std::vector<uint32_t> mol0;    // first global molecule id per molecule type
    std::vector<uint32_t> slot0;   // first accumulator slot per molecule type
    // One read for global leaflets + order parameters (k_bonds_tiled<..., MOM>): when the membrane group is one
    // contiguous range of atoms and the tiles' windows can be made to cover it, tile i OWNS the atoms [own[2i], own[2i+1])
    // — the ranges partition the group — and sums their normal coordinates on the way.
    bool spec_ok = false;
    std::vector<uint32_t> own;
    // the molecules whose HEAD atom a tile owns: (window-relative atom, molecule) per tile (CSR) — the order kernel hands
    // their normal coordinates on, so that the sides can be checked without reading every head's cache line again
    std::vector<uint32_t> own_head_begin;
    std::vector<uint32_t> own_heads;      // pairs
};

struct Sample {
    uint32_t i, j, slot, mol;
};
struct UaSample {
    uint32_t a[4];
    uint32_t slot0, kind, mol, type;
};

// Which lane of the workgroup evaluates which sample of a tile is free (the samples read their atoms from the LDS
// window, not from memory), and it decides the LDS bank conflicts of K1's per-sample reads: a `ds_read_b32` is served
// in two groups of 32 lanes over 32 banks (bank = dword address mod 32), a sample reads w[3 * atom + c], and
// 3 (a - b) = 0 mod 32 only for a = b mod 32 — so two lanes of one half-wave collide exactly when their atoms differ by
// a multiple of 32 (equal atoms broadcast).  Lanes in atom order put 35-50 consecutive atoms on every half-wave, and
// with them such pairs (measured in round 1: 21 % of the LDS cycles on the all-atom, 37 % on the coarse-grained
// membranes).  A conflict-free order does not exist for 256 samples on a window of more than 256 atoms — a half-wave
// of 32 samples touches more than 32 different atoms —, so this is a greedy repair: hand the samples, in atom order,
// to the half-wave in which they meet the fewest atoms of their residue classes.  It is kept only where the modelled
// conflict cycles drop by a fifth or more (chains of beads: 0.50 -> 0.35 of the read cycles, measured 37 % -> 25 %);
// for the carbon-hydrogen fans of an all-atom lipid the model gains < 10 % and the counters got worse (21 % -> 29 %),
// so those tiles stay in atom order.
inline uint32_t bank_conflict_cycles(const Item *items, uint32_t n) {       // extra LDS cycles of one frame's reads
    uint32_t extra = 0;
    for (uint32_t h = 0; h < n; h += 32u)
        for (int which = 0; which < 2; which++) {
            uint16_t seen[32][8];
            uint32_t cnt[32] = {0};
            uint32_t worst = 1;
            for (uint32_t q = h; q < std::min(n, h + 32u); q++) {
                const uint16_t at = which ? items[q].lj : items[q].li;
                uint32_t &c = cnt[at % 32u];
                bool dup = false;
                for (uint32_t k = 0; k < std::min(c, 8u); k++) dup = dup || seen[at % 32u][k] == at;
                if (!dup) { if (c < 8u) seen[at % 32u][c] = at; c++; }
                worst = std::max(worst, c);
            }
            extra += worst - 1u;
        }
    return extra;
}
inline void spread_over_banks(Item *items, uint32_t n) {
    constexpr uint32_t kHalf = 32;
    if (n <= kHalf) return;
    const uint32_t n_bins = (n + kHalf - 1) / kHalf;      // the tile's lanes 0 .. n-1 are the active ones
    auto cap = [&](uint32_t b) { return b + 1 < n_bins ? kHalf : n - kHalf * (n_bins - 1); };
    struct Bin {
        std::vector<uint16_t> at_i[kHalf], at_j[kHalf];     // residue -> the atoms of that class in this half-wave
        std::vector<Item> got;
    };
    std::vector<Bin> bins(n_bins);
    auto added = [](const std::vector<uint16_t> &v, uint16_t a) {
        return (!v.empty() && std::find(v.begin(), v.end(), a) == v.end()) ? 1u : 0u;
    };
    for (uint32_t q = 0; q < n; q++) {
        const Item it = items[q];
        uint32_t best = n_bins, best_cost = ~0u;
        size_t best_fill = 0;
        for (uint32_t b = 0; b < n_bins; b++) {
            Bin &bin = bins[b];
            if (bin.got.size() >= cap(b)) continue;
            const uint32_t cost = added(bin.at_i[it.li % kHalf], it.li) + added(bin.at_j[it.lj % kHalf], it.lj);
            // fewest new collisions, then the fullest half-wave (keeps neighbours together)
            if (cost < best_cost || (cost == best_cost && bin.got.size() > best_fill)) {
                best = b; best_cost = cost; best_fill = bin.got.size();
            }
        }
        Bin &bin = bins[best];
        auto note = [](std::vector<uint16_t> &v, uint16_t a) { if (std::find(v.begin(), v.end(), a) == v.end()) v.push_back(a); };
        note(bin.at_i[it.li % kHalf], it.li);
        note(bin.at_j[it.lj % kHalf], it.lj);
        bin.got.push_back(it);
    }
    std::vector<Item> order;
    for (uint32_t b = 0; b < n_bins; b++) order.insert(order.end(), bins[b].got.begin(), bins[b].got.end());
    if (5u * bank_conflict_cycles(order.data(), n) <= 4u * bank_conflict_cycles(items, n)) std::copy(order.begin(), order.end(), items);
}

// Returns GORDER_OK or GORDER_ERR_INVALID_ARGUMENT (index out of range, self bond, ...).
inline int build_plan(const gorder_tables_t &t, bool force_direct, Plan &p, bool ua_quads = true) {
    p = Plan();
    p.n_atoms = t.n_atoms;
    if (t.n_atoms == 0 || (t.n_molecule_types && !t.molecule_types)) return GORDER_ERR_INVALID_ARGUMENT;
    std::vector<Sample> samples;
    std::vector<UaSample> ua_samples;
    uint32_t slot = 0, mol = 0;
    for (uint32_t m = 0; m < t.n_molecule_types; m++) {
        const gorder_moltype_t &mt = t.molecule_types[m];
        p.mol0.push_back(mol);
        p.slot0.push_back(slot);
        if (mt.n_bond_types && !mt.bonds) return GORDER_ERR_INVALID_ARGUMENT;
        for (uint32_t bt = 0; bt < mt.n_bond_types; bt++) {
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                const uint32_t *b = mt.bonds + 2 * ((size_t)bt * mt.n_molecules + k);
                if (b[0] >= t.n_atoms || b[1] >= t.n_atoms || b[0] == b[1]) return GORDER_ERR_INVALID_ARGUMENT;
                samples.push_back({b[0], b[1], slot + bt, mol + k});
            }
        }
        slot += mt.n_bond_types;
        if (mt.n_ua_atoms && !mt.ua_atoms) return GORDER_ERR_INVALID_ARGUMENT;
        for (uint32_t a = 0; a < mt.n_ua_atoms; a++) {
            const uint32_t kind = mt.ua_atoms[a].kind;
            if (kind < GORDER_UA_CH1_SAT || kind > GORDER_UA_CH1_UNSAT || !mt.ua_atoms[a].indices)
                return GORDER_ERR_INVALID_ARGUMENT;
            const uint32_t nidx = kind == GORDER_UA_CH1_SAT ? 4u : 3u;
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                const uint32_t *ix = mt.ua_atoms[a].indices + 4 * (size_t)k;
                UaSample us{{ix[0], ix[1], ix[2], nidx == 4 ? ix[3] : ix[1]}, slot, kind, mol + k, m};
                for (uint32_t q = 0; q < nidx; q++)
                    if (ix[q] >= t.n_atoms) return GORDER_ERR_INVALID_ARGUMENT;
                ua_samples.push_back(us);
            }
            slot += kind == GORDER_UA_CH3 ? 3 : kind == GORDER_UA_CH2 ? 2 : 1;
        }
        mol += mt.n_molecules;
    }
    p.n_acc = slot;
    p.n_mol_total = mol;

    {   // ---- united-atom tiles
        auto ulo = [](const UaSample &u) { return std::min(std::min(u.a[0], u.a[1]), std::min(u.a[2], u.a[3])); };
        auto uhi = [](const UaSample &u) { return std::max(std::max(u.a[0], u.a[1]), std::max(u.a[2], u.a[3])); };
        // by molecule type first: the molecules of two types may alternate in the atom order, a group is one type's
        std::stable_sort(ua_samples.begin(), ua_samples.end(), [&](const UaSample &x, const UaSample &y) {
            if (x.type != y.type) return x.type < y.type;
            if (ulo(x) != ulo(y)) return ulo(x) < ulo(y);
            return uhi(x) < uhi(y);
        });
        std::vector<std::pair<uint32_t, MapRun>> run_of_slot;
        // One tile = kBlock consecutive samples of [b, e), which the caller has put into lane order.
        auto emit_tile = [&](size_t b, size_t e) {
            Tile tile{};
            tile.atom0 = ulo(ua_samples[b]);
            uint32_t top = 0;
            for (size_t i = b; i < e; i++) { tile.atom0 = std::min(tile.atom0, ulo(ua_samples[i])); top = std::max(top, uhi(ua_samples[i])); }
            tile.item0 = (uint32_t)p.ua_items.size();
            tile.slot0 = (uint32_t)p.ua_tile_slots.size();
            std::vector<uint32_t> slots;   // local slot list holds the FIRST slot of each carbon; 3 entries reserved
            for (size_t i = b; i < e; i++) {
                const UaSample &u = ua_samples[i];
                const uint32_t nh = u.kind == GORDER_UA_CH3 ? 3 : u.kind == GORDER_UA_CH2 ? 2 : 1;
                uint32_t ls = 0;
                for (; ls < slots.size(); ls++)
                    if (slots[ls] == u.slot0) break;
                if (ls == slots.size())
                    for (uint32_t h = 0; h < nh; h++) slots.push_back(u.slot0 + h);
                UaItem it{};
                for (int c = 0; c < 4; c++) it.l[c] = (uint16_t)(u.a[c] - tile.atom0);
                it.lslot0 = (uint16_t)ls;
                it.kind = (uint16_t)u.kind;
                it.mol = u.mol;
                p.ua_items.push_back(it);
                tile.n_items++;
            }
            tile.n_window = top - tile.atom0 + 1;
            tile.n_slots = (uint32_t)slots.size();
            const uint32_t tile_id = (uint32_t)p.ua_tiles.size();
            for (uint32_t i = 0; i < tile.n_items;) {   // the lanes of one slot: a MapRun per hydrogen
                const UaItem &it = p.ua_items[tile.item0 + i];
                uint32_t n = 1;
                while (i + n < tile.n_items && p.ua_items[tile.item0 + i + n].lslot0 == it.lslot0) n++;
                const uint32_t nh = it.kind == GORDER_UA_CH3 ? 3 : it.kind == GORDER_UA_CH2 ? 2 : 1;
                for (uint32_t k = 0; k < nh; k++) run_of_slot.push_back({slots[it.lslot0 + k], MapRun{tile_id, i, n, k}});
                p.ua_item_run.resize((size_t)tile.item0 + tile.n_items, 0u);
                for (uint32_t j = 0; j < n; j++) p.ua_item_run[tile.item0 + i + j] = (i << 16) | n;
                i += n;
            }
            p.ua_tile_slots.insert(p.ua_tile_slots.end(), slots.begin(), slots.end());
            p.ua_tiles.push_back(tile);
        };
        // Tiles are cut from GROUPS of up to 64 molecules (consecutive in atom order): inside a group the carbons go
        // by (kind, slot, molecule), so a wave of 64 lanes holds one kind of carbon — the hydrogen construction
        // branches on the kind, and a wave that mixes methyl, methylene and methine carbons runs all three — and
        // mostly one slot (long MapRuns for the staged ordermap samples).  The window of a tile is then the group's
        // atoms (the kernel gathers through L1 / L2; offsets are 16-bit).
        constexpr uint32_t kUaMaxWindow = 60000, kUaGroupMolecules = 64;
        size_t q = 0;
        while (q < ua_samples.size()) {
            uint32_t span = kUaGroupMolecules;
            size_t e = q;
            for (;;) {
                const uint32_t mol0 = ua_samples[q].mol;
                uint32_t lo_atom = ulo(ua_samples[q]), hi_atom = uhi(ua_samples[q]);
                for (e = q; e < ua_samples.size(); e++) {
                    const UaSample &u = ua_samples[e];
                    if (u.type != ua_samples[q].type || u.mol < mol0 || u.mol - mol0 >= span) break;
                    lo_atom = std::min(lo_atom, ulo(u));
                    hi_atom = std::max(hi_atom, uhi(u));
                }
                if (hi_atom - lo_atom + 1 <= kUaMaxWindow) break;
                if (span == 1) return GORDER_ERR_INVALID_ARGUMENT;      // one molecule wider than 16-bit offsets reach
                span /= 2;
            }
            if (!ua_quads) {
                std::stable_sort(ua_samples.begin() + q, ua_samples.begin() + e, [](const UaSample &x, const UaSample &y) {
                    if (x.kind != y.kind) return x.kind < y.kind;
                    if (x.slot0 != y.slot0) return x.slot0 < y.slot0;
                    return x.mol < y.mol;
                });
            } else {
                // A wave = 4 neighbouring carbons (slots) of one kind x 16 molecules, a DPP row of 16 lanes per slot: the
                // lanes of a wave then gather from 16 molecules' neighbourhoods (a line or two each) instead of from 64
                // molecules (64 lines per load instruction), and a slot's lanes are still a run of 16 for the staged
                // ordermap samples and the per-frame sums.
                std::map<std::pair<uint32_t, uint32_t>, uint32_t> rank;      // (kind, slot0) -> rank of the slot in its kind
                {
                    std::vector<std::pair<uint32_t, uint32_t>> ks;
                    for (size_t i = q; i < e; i++) ks.push_back({ua_samples[i].kind, ua_samples[i].slot0});
                    std::sort(ks.begin(), ks.end());
                    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
                    uint32_t r = 0;
                    for (size_t i = 0; i < ks.size(); i++) {
                        if (i && ks[i].first != ks[i - 1].first) r = 0;
                        rank[ks[i]] = r++;
                    }
                }
                const uint32_t mol_first = ua_samples[q].mol;
                auto key = [&](const UaSample &u) {
                    const uint32_t r = rank[{u.kind, u.slot0}];
                    return std::make_tuple(u.kind, r / 4u, (u.mol - mol_first) / 16u, u.slot0, u.mol);
                };
                std::stable_sort(ua_samples.begin() + q, ua_samples.begin() + e,
                                 [&](const UaSample &x, const UaSample &y) { return key(x) < key(y); });
            }
            for (size_t b = q; b < e; b += kBlock) emit_tile(b, std::min(e, b + kBlock));
            q = e;
        }
        std::stable_sort(run_of_slot.begin(), run_of_slot.end(),
                         [](const std::pair<uint32_t, MapRun> &x, const std::pair<uint32_t, MapRun> &y) { return x.first < y.first; });
        p.ua_run_begin.assign((size_t)p.n_acc + 1, 0);
        for (const auto &r : run_of_slot) { p.ua_run_begin[r.first + 1]++; p.ua_runs.push_back(r.second); }
        for (uint32_t sl = 0; sl < p.n_acc; sl++) p.ua_run_begin[sl + 1] += p.ua_run_begin[sl];
    }

    auto lo = [](const Sample &s) { return std::min(s.i, s.j); };
    auto hi = [](const Sample &s) { return std::max(s.i, s.j); };
    std::stable_sort(samples.begin(), samples.end(), [&](const Sample &a, const Sample &b) {
        if (lo(a) != lo(b)) return lo(a) < lo(b);
        return hi(a) < hi(b);
    });

    size_t k = 0;
    while (k < samples.size()) {
        const Sample &first = samples[k];
        if (force_direct || hi(first) - lo(first) + 1 > kMaxWindow) {
            p.direct.push_back({first.i, first.j, first.slot, first.mol});
            k++;
            continue;
        }
        Tile tile{};
        tile.atom0 = lo(first);
        tile.item0 = (uint32_t)p.items.size();
        tile.slot0 = (uint32_t)p.tile_slots.size();
        uint32_t top = hi(first);
        std::vector<uint32_t> slots;
        while (k < samples.size() && tile.n_items < kBlock) {
            const Sample &s = samples[k];
            if (hi(s) - lo(s) + 1 > kMaxWindow) {   // oversized sample: direct list, keep packing
                p.direct.push_back({s.i, s.j, s.slot, s.mol});
                k++;
                continue;
            }
            const uint32_t ntop = std::max(top, hi(s));
            if (ntop - tile.atom0 + 1 > kMaxWindow) break;
            top = ntop;
            uint32_t ls = 0;
            for (; ls < slots.size(); ls++)
                if (slots[ls] == s.slot) break;
            if (ls == slots.size()) slots.push_back(s.slot);
            p.items.push_back({(uint16_t)(s.i - tile.atom0), (uint16_t)(s.j - tile.atom0), (uint16_t)ls, 0, s.mol});
            tile.n_items++;
            k++;
        }
        tile.n_window = top - tile.atom0 + 1;
        tile.n_slots = (uint32_t)slots.size();
        p.tile_slots.insert(p.tile_slots.end(), slots.begin(), slots.end());
        p.max_window = std::max(p.max_window, tile.n_window);
        spread_over_banks(p.items.data() + tile.item0, tile.n_items);
        p.tiles.push_back(tile);
    }
    // ---- ownership ranges of the membrane group's atoms (see Plan::own)
    p.spec_ok = false;
    p.own.clear();
    if (t.leaflets.method == GORDER_LEAFLETS_GLOBAL && t.leaflets.n_membrane > 0 && t.leaflets.membrane && p.direct.empty() &&
        !p.tiles.empty() && p.ua_tiles.empty()) {
        const uint32_t *mb = t.leaflets.membrane;
        const uint32_t nm = t.leaflets.n_membrane, m0 = mb[0], m1 = m0 + nm;
        bool ok = m1 <= t.n_atoms && m1 > m0;
        for (uint32_t i = 1; ok && i < nm; i++) ok = mb[i] == m0 + i;
        for (size_t i = 1; ok && i < p.tiles.size(); i++) ok = p.tiles[i].atom0 >= p.tiles[i - 1].atom0;
        if (ok) {
            std::vector<Tile> nt = p.tiles;
            std::vector<uint32_t> own(2 * nt.size());
            uint32_t cur = m0, shift0 = 0, widest = 0;
            for (size_t i = 0; ok && i < nt.size(); i++) {
                const uint32_t lo = cur;
                const uint32_t hi = i + 1 < nt.size() ? std::max(cur, std::min(nt[i + 1].atom0, m1)) : m1;
                if (lo < hi) {
                    if (lo < nt[i].atom0) {                     // (the first tile only: the group begins before its first sample)
                        if (i != 0) { ok = false; break; }
                        shift0 = nt[i].atom0 - lo;
                        nt[i].atom0 = lo;
                        nt[i].n_window += shift0;
                    }
                    if (hi > nt[i].atom0 + nt[i].n_window) nt[i].n_window = hi - nt[i].atom0;
                    if (nt[i].n_window > kMaxWindow) ok = false;
                }
                own[2 * i] = lo;
                own[2 * i + 1] = hi;
                cur = hi;
                widest = std::max(widest, nt[i].n_window);
            }
            // every molecule's head must be an atom of the group (then exactly one tile owns it)
            std::vector<uint32_t> head_begin, heads_flat;
            if (ok && cur == m1) {
                std::vector<std::vector<uint32_t>> per_tile(nt.size());
                uint32_t mol = 0;
                for (uint32_t m = 0; ok && m < t.n_molecule_types; m++) {
                    const gorder_moltype_t &mt = t.molecule_types[m];
                    for (uint32_t k = 0; ok && k < mt.n_molecules; k++, mol++) {
                        if (!mt.heads) { ok = false; break; }
                        const uint32_t hd = mt.heads[k];
                        if (hd < m0 || hd >= m1) { ok = false; break; }
                        size_t lo_t = 0, hi_t = nt.size();          // the tile whose [own0, own1) holds hd (ranges ascend)
                        while (hi_t - lo_t > 1) {
                            const size_t mid = (lo_t + hi_t) / 2;
                            if (own[2 * mid] <= hd) lo_t = mid; else hi_t = mid;
                        }
                        while (lo_t > 0 && !(own[2 * lo_t] <= hd && hd < own[2 * lo_t + 1])) lo_t--;    // (empty ranges share a start)
                        if (!(own[2 * lo_t] <= hd && hd < own[2 * lo_t + 1])) { ok = false; break; }
                        per_tile[lo_t].push_back(hd - nt[lo_t].atom0);
                        per_tile[lo_t].push_back(mol);
                    }
                }
                head_begin.push_back(0);
                for (const auto &v : per_tile) {
                    if (v.size() > 2 * 64) ok = false;             // (the order kernel holds a tile's heads one a lane)
                    heads_flat.insert(heads_flat.end(), v.begin(), v.end());
                    head_begin.push_back((uint32_t)(heads_flat.size() / 2));
                }
            }
            if (ok && cur == m1) {
                p.own_head_begin = head_begin;
                p.own_heads = heads_flat;
                if (shift0)
                    for (uint32_t q = 0; q < nt[0].n_items; q++) {
                        p.items[nt[0].item0 + q].li = (uint16_t)(p.items[nt[0].item0 + q].li + shift0);
                        p.items[nt[0].item0 + q].lj = (uint16_t)(p.items[nt[0].item0 + q].lj + shift0);
                    }
                p.tiles = nt;
                p.max_window = std::max(p.max_window, widest);
                p.own = own;
                p.spec_ok = true;
            }
        }
    }
    {   // slot-ordered copy of the tile items + the runs of every slot (see MapRun)
        p.items_by_slot = p.items;
        std::vector<std::pair<uint32_t, MapRun>> run_of_slot;
        for (uint32_t ti = 0; ti < p.tiles.size(); ti++) {
            const Tile &tile = p.tiles[ti];
            std::stable_sort(p.items_by_slot.begin() + tile.item0, p.items_by_slot.begin() + tile.item0 + tile.n_items,
                             [](const Item &x, const Item &y) { return x.lslot < y.lslot; });
            p.item_run.resize((size_t)tile.item0 + tile.n_items, 0u);
            for (uint32_t i = 0; i < tile.n_items;) {
                const uint16_t ls = p.items_by_slot[tile.item0 + i].lslot;
                uint32_t n = 1;
                while (i + n < tile.n_items && p.items_by_slot[tile.item0 + i + n].lslot == ls) n++;
                run_of_slot.push_back({p.tile_slots[tile.slot0 + ls], MapRun{ti, i, n, 0}});
                for (uint32_t j = 0; j < n; j++) p.item_run[tile.item0 + i + j] = (i << 16) | n;
                i += n;
            }
        }
        std::stable_sort(run_of_slot.begin(), run_of_slot.end(),
                         [](const std::pair<uint32_t, MapRun> &x, const std::pair<uint32_t, MapRun> &y) { return x.first < y.first; });
        p.run_begin.assign((size_t)p.n_acc + 1, 0);
        for (const auto &r : run_of_slot) { p.run_begin[r.first + 1]++; p.runs.push_back(r.second); }
        for (uint32_t sl = 0; sl < p.n_acc; sl++) p.run_begin[sl + 1] += p.run_begin[sl];
    }
    return GORDER_OK;
}

// Every sample of the tables appears exactly once (tile item or direct item) with the right atoms,
// slot and molecule; windows stay inside the frame.  Used by the CPU test-suite.
inline int selfcheck_plan(const gorder_tables_t &t, const Plan &p) {
    std::vector<Sample> want, got;
    uint32_t slot = 0, mol = 0;
    for (uint32_t m = 0; m < t.n_molecule_types; m++) {
        const gorder_moltype_t &mt = t.molecule_types[m];
        for (uint32_t bt = 0; bt < mt.n_bond_types; bt++)
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                const uint32_t *b = mt.bonds + 2 * ((size_t)bt * mt.n_molecules + k);
                want.push_back({b[0], b[1], slot + bt, mol + k});
            }
        slot += mt.n_bond_types;
        for (uint32_t a = 0; a < mt.n_ua_atoms; a++) {
            const uint32_t kind = mt.ua_atoms[a].kind;
            slot += kind == GORDER_UA_CH3 ? 3 : kind == GORDER_UA_CH2 ? 2 : 1;
        }
        mol += mt.n_molecules;
    }
    for (const Tile &tile : p.tiles) {
        if (tile.n_items == 0 || tile.n_items > kBlock) return 1;
        if (tile.n_window == 0 || tile.n_window > kMaxWindow) return 2;
        if ((uint64_t)tile.atom0 + tile.n_window > t.n_atoms) return 3;
        if (tile.n_slots == 0 || tile.n_slots > tile.n_items) return 4;
        for (uint32_t q = 0; q < tile.n_items; q++) {
            const Item &it = p.items[tile.item0 + q];
            if (it.li >= tile.n_window || it.lj >= tile.n_window || it.lslot >= tile.n_slots) return 5;
            got.push_back({tile.atom0 + it.li, tile.atom0 + it.lj, p.tile_slots[tile.slot0 + it.lslot], it.mol});
        }
    }
    for (const DirectItem &d : p.direct) got.push_back({d.i, d.j, d.slot, d.mol});
    auto cmp = [](const Sample &a, const Sample &b) {
        if (a.slot != b.slot) return a.slot < b.slot;
        if (a.mol != b.mol) return a.mol < b.mol;
        if (a.i != b.i) return a.i < b.i;
        return a.j < b.j;
    };
    std::sort(want.begin(), want.end(), cmp);
    std::sort(got.begin(), got.end(), cmp);
    if (want.size() != got.size()) return 6;
    for (size_t q = 0; q < want.size(); q++)
        if (want[q].i != got[q].i || want[q].j != got[q].j || want[q].slot != got[q].slot ||
            want[q].mol != got[q].mol)
            return 7;
    // the slot-ordered copy is a permutation of each tile's items, and the runs of every slot cover exactly the
    // lanes that hold that slot (bond tiles: one entry per lane; united-atom tiles: one per lane and hydrogen)
    if (p.items_by_slot.size() != p.items.size() || p.run_begin.size() != (size_t)p.n_acc + 1) return 8;
    std::vector<uint32_t> covered(p.items.size(), 0);
    for (uint32_t sl = 0; sl < p.n_acc; sl++)
        for (uint32_t r = p.run_begin[sl]; r < p.run_begin[sl + 1]; r++) {
            const MapRun &run = p.runs[r];
            if (run.tile >= p.tiles.size() || run.k != 0 || run.tid0 + run.n > p.tiles[run.tile].n_items) return 9;
            const Tile &tile = p.tiles[run.tile];
            for (uint32_t j = 0; j < run.n; j++) {
                const Item &it = p.items_by_slot[tile.item0 + run.tid0 + j];
                if (p.tile_slots[tile.slot0 + it.lslot] != sl) return 10;
                if (p.item_run.size() != p.items.size() || p.item_run[tile.item0 + run.tid0 + j] != ((run.tid0 << 16) | run.n))
                    return 16;
                covered[tile.item0 + run.tid0 + j]++;
            }
        }
    for (uint32_t c : covered)
        if (c != 1) return 11;
    if (!p.ua_tiles.empty()) {
        if (p.ua_run_begin.size() != (size_t)p.n_acc + 1) return 12;
        size_t lanes = 0, want_lanes = 0;
        for (const UaItem &it : p.ua_items) want_lanes += it.kind == GORDER_UA_CH3 ? 3 : it.kind == GORDER_UA_CH2 ? 2 : 1;
        for (uint32_t sl = 0; sl < p.n_acc; sl++)
            for (uint32_t r = p.ua_run_begin[sl]; r < p.ua_run_begin[sl + 1]; r++) {
                const MapRun &run = p.ua_runs[r];
                if (run.tile >= p.ua_tiles.size() || run.k > 2 || run.tid0 + run.n > p.ua_tiles[run.tile].n_items) return 13;
                const Tile &tile = p.ua_tiles[run.tile];
                for (uint32_t j = 0; j < run.n; j++) {
                    const UaItem &it = p.ua_items[tile.item0 + run.tid0 + j];
                    if (p.ua_tile_slots[tile.slot0 + it.lslot0 + run.k] != sl) return 14;
                    if (p.ua_item_run.size() != p.ua_items.size() ||
                        p.ua_item_run[tile.item0 + run.tid0 + j] != ((run.tid0 << 16) | run.n))
                        return 17;
                    lanes++;
                }
            }
        if (lanes != want_lanes) return 15;
    }
    return 0;
}

}  // namespace gorder
