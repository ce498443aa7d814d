// tables_model_test.cpp — the host layer's id-keyed containers (swarmkit_amd/csrc/swp_tables.hpp: IdTable, OrderedTasks, NodeTasks) against
// std::map / a plain vector under long random operation sequences: what they answer, the order they iterate in, and that the holes an
// IdTable closes (more than 1 024 of them and more than half of the entries) and the growth of its index lose nothing.
// TEST INFRASTRUCTURE (built and run by tests/test_host_tables_cpu.py); usage: tables_model_test <seed> <operations>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../swarmkit_amd/csrc/swp_tables.hpp"

using swp::IdTable;
using swp::NodeTasks;
using swp::OrderedTasks;
using swp::json::Value;

static int bad = 0;
#define CHECK(c) do { if (!(c)) { if (bad++ < 10) std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

struct Payload { int v = 0; bool flag = false; };

int main(int argc, char** argv) {
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    const long ops = argc > 2 ? std::atol(argv[2]) : 200000;
    std::mt19937_64 rng(seed);
    auto id_of = [&](unsigned universe) { return "task-" + std::to_string(rng() % universe); };

    // ---- IdTable: most of a large table erased one by one (the holes are closed on the way), then filled again ----
    {
        IdTable<Payload> t;
        std::map<std::string, Payload> m;
        for (int i = 0; i < 6000; ++i) { t["k" + std::to_string(i)] = Payload{i, false}; m["k" + std::to_string(i)] = Payload{i, false}; }
        std::vector<int> order(6000);
        for (int i = 0; i < 6000; ++i) order[(size_t)i] = i;
        std::shuffle(order.begin(), order.end(), rng);
        for (int j = 0; j < 5500; ++j) {
            const std::string id = "k" + std::to_string(order[(size_t)j]);
            CHECK(t.erase(id));
            m.erase(id);
            if (j % 500 == 0)
                for (const auto& kv : m) { auto* e = t.find(kv.first); CHECK(e != t.end() && e->second.v == kv.second.v); }
        }
        CHECK(t.size() == m.size());
        for (const auto& kv : m) { auto* e = t.find(kv.first); CHECK(e != t.end() && e->second.v == kv.second.v); }
        for (int j = 0; j < 5500; ++j) CHECK(t.find("k" + std::to_string(order[(size_t)j])) == t.end());
        for (int i = 0; i < 3000; ++i) t["n" + std::to_string(i)] = Payload{i, true};
        CHECK(t.size() == m.size() + 3000);
        auto s = t.sorted();
        for (size_t k = 1; k < s.size(); ++k) CHECK(s[k - 1]->first < s[k]->first);
    }
    // ---- IdTable against std::map ----
    {
        IdTable<Payload> t;
        std::map<std::string, Payload> m;
        unsigned universe = 50;
        for (long i = 0; i < ops; ++i) {
            if (i % 20000 == 0) universe = (unsigned)(rng() % 3 == 0 ? 40 : (rng() % 2 ? 3000 : 20000));   // phases: tiny, medium, large key spaces
            const unsigned r = (unsigned)(rng() % 100);
            const std::string id = id_of(universe);
            if (r < 45) {
                const int v = (int)(rng() % 1000);
                if (rng() % 2) t[id] = Payload{v, v % 3 == 0};
                else t.at(id, swp::id_hash(id)) = Payload{v, v % 3 == 0};
                m[id] = Payload{v, v % 3 == 0};
            } else if (r < 75) {
                CHECK(t.erase(id) == (m.erase(id) != 0));
            } else if (r < 93) {
                auto* e = rng() % 2 ? t.find(id) : t.find(id, swp::id_hash(id));
                auto it = m.find(id);
                CHECK((e != t.end()) == (it != m.end()));
                if (e != t.end() && it != m.end()) CHECK(e->first == id && e->second.v == it->second.v && e->second.flag == it->second.flag);
            } else if (r < 95) {
                const bool which = rng() % 2;
                t.erase_if([&](const Payload& p) { return p.flag == which; });
                for (auto it = m.begin(); it != m.end();) it = it->second.flag == which ? m.erase(it) : std::next(it);
            } else if (r < 97) {
                t.reserve(t.size() + (size_t)(rng() % 5000));
            } else if (r < 98 && rng() % 4 == 0) {
                t.clear();
                m.clear();
            } else {
                auto s = t.sorted();
                CHECK(s.size() == m.size());
                size_t k = 0;
                for (const auto& kv : m) {
                    if (k < s.size()) CHECK(s[k]->first == kv.first && s[k]->second.v == kv.second.v);
                    ++k;
                }
            }
            CHECK(t.size() == m.size() && t.empty() == m.empty());
            t.prefetch(swp::id_hash(id), 1);
            t.prefetch(swp::id_hash(id), 2);
        }
    }
    // ---- OrderedTasks: insertion order, assignment keeps the position, take_all empties ----
    {
        OrderedTasks q;
        std::vector<std::pair<std::string, long>> order;   // id, value (alive ones, in first-insertion order)
        for (long i = 0; i < ops / 4; ++i) {
            const unsigned r = (unsigned)(rng() % 100);
            const std::string id = id_of(2000);
            if (r < 60) {
                q.put(id, Value::integer(i), (uint32_t)(i % 7));
                bool found = false;
                for (auto& kv : order)
                    if (kv.first == id) { kv.second = i; found = true; }
                if (!found) order.emplace_back(id, i);
            } else if (r < 90) {
                q.erase(id);
                for (size_t k = 0; k < order.size(); ++k)
                    if (order[k].first == id) { order.erase(order.begin() + (long)k); break; }
            } else if (r < 95) {
                auto snap = q.snapshot();
                CHECK(snap.size() == order.size());
                for (size_t k = 0; k < snap.size() && k < order.size(); ++k) CHECK(snap[k].first == order[k].first && snap[k].second.i == order[k].second);
            } else {
                auto all = q.take_all();
                CHECK(all.size() == order.size());
                for (size_t k = 0; k < all.size() && k < order.size(); ++k) CHECK(all[k].first == order[k].first && all[k].second.i == order[k].second);
                CHECK(q.empty() && q.size() == 0);
                order.clear();
            }
            CHECK(q.size() == order.size());
        }
    }
    // ---- OrderedTasks that is never taken whole (the preassigned queue: ONE task that never fits keeps it alive): the erased entries
    // of everything that came and went must not pile up (300k put / erase pairs next to one live entry held 49 MB) ----
    {
        OrderedTasks q;
        q.put("stays", Value::integer(-1));
        std::vector<std::string> live{"stays"};
        size_t worst = 0;
        for (long i = 0; i < 300000; ++i) {
            const std::string id = "t" + std::to_string(i);
            q.put(id, Value::integer(i));
            if (i % 1000 == 999) live.push_back(id);   // a few more stay for good
            else q.erase(id);
            worst = std::max(worst, q.slots());
            if (i % 50000 == 0) {
                auto snap = q.snapshot();
                CHECK(snap.size() == live.size());
                for (size_t k = 0; k < snap.size() && k < live.size(); ++k) CHECK(snap[k].first == live[k]);
            }
        }
        CHECK(q.size() == live.size());
        CHECK(worst <= 2 * live.size() + 1100);   // holes never outnumber the entries by more than the compaction's slack
        auto all = q.take_all();
        CHECK(all.size() == live.size());
        for (size_t k = 0; k < all.size() && k < live.size(); ++k) CHECK(all[k].first == live[k]);
    }
    // ---- NodeTasks: flat while small, a tree beyond 48 entries; sorted iteration ----
    {
        for (int round = 0; round < 200; ++round) {
            NodeTasks nt;
            std::map<std::string, long> m;
            const unsigned universe = round % 2 ? 30 : 200;   // (stays flat / goes over to the tree)
            for (long i = 0; i < ops / 400; ++i) {
                const unsigned r = (unsigned)(rng() % 100);
                const std::string id = id_of(universe);
                if (r < 55) {
                    if (rng() % 2) nt.put(id, Value::integer(i));
                    else nt.put(id, swp::id_hash(id), Value::integer(i));
                    m[id] = i;
                } else if (r < 80) {
                    CHECK(nt.erase(id) == (m.erase(id) != 0));
                } else if (r < 95) {
                    Value* v = nt.find(id);
                    auto it = m.find(id);
                    CHECK((v != nullptr) == (it != m.end()));
                    if (v != nullptr && it != m.end()) CHECK(v->i == it->second);
                } else {
                    std::vector<std::pair<std::string, long>> got;
                    nt.each_sorted([&](const std::string& k, const Value& v) { got.emplace_back(k, v.i); });
                    CHECK(got.size() == m.size());
                    size_t k = 0;
                    for (const auto& kv : m) {
                        if (k < got.size()) CHECK(got[k].first == kv.first && got[k].second == kv.second);
                        ++k;
                    }
                }
                nt.prefetch();
            }
        }
    }
    std::fprintf(stderr, "tables_model_test seed %u, %ld operations -> %s\n", seed, ops, bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}
