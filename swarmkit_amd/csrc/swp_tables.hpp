// swp_tables.hpp — the id-keyed containers of the host layer (swp_sched.cpp): the tables a tick goes through task by task. They stand in
// for the Go maps of manager/scheduler (Scheduler.allTasks / unassignedTasks / pendingPreassignedTasks, scheduler.go:37-41; NodeInfo.Tasks,
// nodeinfo.go:31) with the properties a 100k-task tick needs: no node per entry, insertion order where the reference's behaviour depends
// on none (Go ranges over maps in random order; the canonical order here is first insertion, the oracle's), one hash of an id for all of
// them.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "swp_json.hpp"

namespace swp {

using json::Value;

// An insertion-ordered map (Go code ranges over maps in random order; the canonical order of this implementation is
// first-insertion order, the same as the oracle's): assigning an existing key keeps its position.

// One queued task: its id, its document, and the TEMPLATE it was recognised as when the event came in (Scheduler::templateOf: the tasks of
// a service share everything Pipeline.SetTask reads; NO_TMPL: not looked up yet). Named like a pair: the handlers below read .first / .second.
static constexpr uint32_t NO_TMPL = 0xFFFFFFFFu;
struct QItem {
    std::string first;
    Value second;
    uint32_t tmpl = NO_TMPL;
    QItem() = default;
    QItem(std::string id, Value t, uint32_t tm = NO_TMPL) : first(std::move(id)), second(std::move(t)), tmpl(tm) {}
};

// One hash of a task id serves every table a decision touches (allTasks, the decision log, NodeInfo.Tasks).
static inline uint64_t id_hash(const std::string& s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (unsigned char c : s) h = (h ^ c) * 0x100000001B3ull;
    h ^= h >> 29;
    return h ? h : 1;
}

// id -> V for the maps a tick goes through task by task (Scheduler.allTasks, the decisions of the last tick): entries in ONE vector in
// insertion order, found through an open-addressing table of (hash, position) pairs. Against std::unordered_map / std::map this is no
// node per entry (a 100k-task tick: 100k allocations less per map, and dropping the decisions of the previous tick is one fill), one
// probe sequence over 12 bytes per cell instead of a chain walk, and entries that were created together lie together.
// find() returns a pointer (end() is nullptr) that stays valid until the next put / operator[] / erase.
template <class V> class IdTable {
  public:
    struct Entry { std::string first; V second; };
    Entry* end() const { return nullptr; }
    Entry* find(const std::string& id) { return find(id, id_hash(id)); }
    Entry* find(const std::string& id, uint64_t h) {
        const size_t at = probe(id, h);
        return at == NPOS ? nullptr : &items_[slot_[at]];
    }
    const Entry* find(const std::string& id) const { return const_cast<IdTable*>(this)->find(id); }
    V& operator[](const std::string& id) { return at(id, id_hash(id)); }
    V& at(const std::string& id, uint64_t h) {   // the entry of id, made if absent
        const size_t c = probe(id, h);
        if (c != NPOS) return items_[slot_[c]].second;
        if ((used_ + 1) * 2 > hash_.size()) grow(live_ + 1);
        insert(h, (uint32_t)items_.size());
        items_.push_back(Entry{id, V()});
        alive_.push_back(1);
        ++live_;
        return items_.back().second;
    }
    bool erase(const std::string& id) {
        const size_t c = probe(id, id_hash(id));
        if (c == NPOS) return false;
        alive_[slot_[c]] = 0;
        items_[slot_[c]] = Entry();
        slot_[c] = GONE;   // (the probe chain stays intact)
        if (--live_ == 0) clear();
        else if (items_.size() - live_ > 1024 && items_.size() > 2 * live_) grow(live_);   // mostly holes: close them
        return true;
    }
    template <class P> void erase_if(P pred) {
        size_t hit = 0;
        for (size_t i = 0; i < items_.size(); ++i) hit += alive_[i] && pred(items_[i].second) ? 1 : 0;
        if (hit == 0) return;
        if (hit == live_) { clear(); return; }
        for (size_t i = 0; i < items_.size(); ++i)
            if (alive_[i] && pred(items_[i].second)) {
                alive_[i] = 0;
                items_[i] = Entry();
                --live_;
            }
        grow(live_);   // rebuilds the index from the entries that are left
    }
    void clear() {
        items_.clear();
        alive_.clear();
        std::fill(hash_.begin(), hash_.end(), 0);
        used_ = live_ = 0;
    }
    size_t size() const { return live_; }
    bool empty() const { return live_ == 0; }
    // What a look-up of h will read, asked for ahead of time (a tick knows the ids it is going to book): 1 = the table cell, 2 = the
    // entry the cell names (once the cell is there).
    void prefetch(uint64_t h, int stage) const {
        if (hash_.empty()) return;
        const size_t i = (size_t)h & (hash_.size() - 1);
        if (stage == 1) {
            __builtin_prefetch(&hash_[i]);
            __builtin_prefetch(&slot_[i]);
        } else if (hash_[i] == h && slot_[i] < items_.size()) __builtin_prefetch(&items_[slot_[i]]);
    }
    void reserve(size_t n) {   // room for n entries without a move of the entries or a rebuild of the table on the way
        if (hash_.size() < 2 * n) grow(n);
        items_.reserve(n);
        alive_.reserve(n);
    }
    // the entries in ascending id order (what ranging over a std::map gave: deterministic outputs)
    std::vector<const Entry*> sorted() const {
        std::vector<const Entry*> out;
        out.reserve(live_);
        bool ordered = true;
        for (size_t i = 0; i < items_.size(); ++i)
            if (alive_[i]) {
                ordered = ordered && (out.empty() || out.back()->first < items_[i].first);
                out.push_back(&items_[i]);
            }
        if (!ordered) std::sort(out.begin(), out.end(), [](const Entry* a, const Entry* b) { return a->first < b->first; });
        return out;
    }

  private:
    static constexpr size_t NPOS = ~(size_t)0;
    static constexpr uint32_t GONE = 0xFFFFFFFFu;
    size_t probe(const std::string& id, uint64_t h) const {
        if (hash_.empty()) return NPOS;
        const size_t mask = hash_.size() - 1;
        for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
            if (hash_[i] == 0) return NPOS;
            if (hash_[i] == h && slot_[i] != GONE && items_[slot_[i]].first == id) return i;
        }
    }
    void insert(uint64_t h, uint32_t pos) {
        const size_t mask = hash_.size() - 1;
        size_t i = (size_t)h & mask;
        while (hash_[i] != 0) i = (i + 1) & mask;
        hash_[i] = h;
        slot_[i] = pos;
        ++used_;
    }
    // a table for at least `want` entries, the holes of items_ closed
    void grow(size_t want) {
        if (items_.size() != live_) {
            std::vector<Entry> kept;
            kept.reserve(std::max(live_, want));
            for (size_t i = 0; i < items_.size(); ++i)
                if (alive_[i]) kept.push_back(std::move(items_[i]));
            items_.swap(kept);
            alive_.assign(items_.size(), 1);
        }
        size_t cap = 1024;
        while (cap < 4 * want) cap *= 2;
        hash_.assign(cap, 0);
        slot_.assign(cap, 0);
        used_ = 0;
        for (size_t i = 0; i < items_.size(); ++i) insert(id_hash(items_[i].first), (uint32_t)i);
    }
    std::vector<Entry> items_;
    std::vector<char> alive_;
    std::vector<uint64_t> hash_;
    std::vector<uint32_t> slot_;
    size_t used_ = 0;   // cells taken (erased ones included)
    size_t live_ = 0;
};

class OrderedTasks {   // (a vector with tombstones: a tick takes the whole queue at once; a queue that is never taken whole — the preassigned
                       // tasks, where one task that never fits keeps it alive — closes its holes once they outnumber the entries: compact())
  public:
    void put(const std::string& id, Value t, uint32_t tmpl = NO_TMPL) {
        const uint64_t h = hash_id(id);
        const size_t at = probe(id, h);
        if (at != NPOS) {
            items_[slot_[at]].second = std::move(t);
            items_[slot_[at]].tmpl = tmpl;
            return;
        }
        if ((used_ + 1) * 2 > hash_.size()) grow();
        insert(h, (uint32_t)items_.size());
        items_.emplace_back(id, std::move(t), tmpl);
        alive_.push_back(1);
        ++live_;
    }
    const Value* find(const std::string& id) const {   // the queued document, nullptr: the id is not queued
        const size_t at = probe(id, hash_id(id));
        return at == NPOS ? nullptr : &items_[slot_[at]].second;
    }
    void erase(const std::string& id) {
        const size_t at = probe(id, hash_id(id));
        if (at == NPOS) return;
        alive_[slot_[at]] = 0;
        items_[slot_[at]].second = Value();
        slot_[at] = GONE;   // (the probe chain stays intact)
        if (--live_ == 0) clear();
        else if (items_.size() - live_ > 1024 && items_.size() > 2 * live_) compact();
    }
    void clear() {
        items_.clear();
        alive_.clear();
        std::fill(hash_.begin(), hash_.end(), 0);
        used_ = live_ = 0;
    }
    // the live entries in their order, the erased ones dropped; the index rebuilt for what is left (sized from live_, not doubled)
    void compact() {
        std::vector<QItem> kept;
        kept.reserve(live_);
        for (size_t i = 0; i < items_.size(); ++i)
            if (alive_[i]) kept.push_back(std::move(items_[i]));
        items_.swap(kept);
        alive_.assign(items_.size(), 1);
        size_t cap = 1024;
        while (cap < 4 * items_.size()) cap *= 2;
        hash_.assign(cap, 0);
        slot_.assign(cap, 0);
        used_ = 0;
        for (size_t i = 0; i < items_.size(); ++i) insert(hash_id(items_[i].first), (uint32_t)i);
    }
    size_t slots() const { return items_.size(); }   // entries held, erased ones included (tests)
    std::vector<QItem> snapshot() const {
        std::vector<QItem> out;
        out.reserve(live_);
        for (size_t i = 0; i < items_.size(); ++i)
            if (alive_[i]) out.push_back(items_[i]);
        return out;
    }
    // the queue's content in order, MOVED out (the queue is empty afterwards): a tick takes everything
    std::vector<QItem> take_all() {
        std::vector<QItem> out;
        if (live_ == items_.size()) out = std::move(items_);
        else {
            out.reserve(live_);
            for (size_t i = 0; i < items_.size(); ++i)
                if (alive_[i]) out.push_back(std::move(items_[i]));
        }
        clear();
        return out;
    }
    bool empty() const { return live_ == 0; }
    size_t size() const { return live_; }

  private:
    // id -> position in items_: open addressing over (hash, position) pairs — no node per entry, so emptying the queue of a 100k-task
    // tick is one fill instead of 100k frees. hash 0 = a free cell; position GONE = erased (the chain goes on).
    static constexpr size_t NPOS = ~(size_t)0;
    static constexpr uint32_t GONE = 0xFFFFFFFFu;
    static uint64_t hash_id(const std::string& s) { return id_hash(s); }
    size_t probe(const std::string& id, uint64_t h) const {
        if (hash_.empty()) return NPOS;
        const size_t mask = hash_.size() - 1;
        for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
            if (hash_[i] == 0) return NPOS;
            if (hash_[i] == h && slot_[i] != GONE && items_[slot_[i]].first == id) return i;
        }
    }
    void insert(uint64_t h, uint32_t pos) {
        const size_t mask = hash_.size() - 1;
        size_t i = (size_t)h & mask;
        while (hash_[i] != 0) i = (i + 1) & mask;
        hash_[i] = h;
        slot_[i] = pos;
        ++used_;
    }
    void grow() {
        if (items_.size() - live_ > 1024 && items_.size() > 2 * live_) { compact(); if ((used_ + 1) * 2 <= hash_.size()) return; }   // (cells of erased entries: not a reason to double)
        std::vector<uint64_t> oh;
        std::vector<uint32_t> os;
        oh.swap(hash_);
        os.swap(slot_);
        const size_t cap = oh.empty() ? 1024 : oh.size() * 2;
        hash_.assign(cap, 0);
        slot_.assign(cap, 0);
        used_ = 0;
        for (size_t i = 0; i < oh.size(); ++i)
            if (oh[i] != 0 && os[i] != GONE) insert(oh[i], os[i]);
    }
    std::vector<QItem> items_;
    std::vector<char> alive_;
    std::vector<uint64_t> hash_;
    std::vector<uint32_t> slot_;
    size_t used_ = 0;   // cells taken (erased ones included)
    size_t live_ = 0;
};

// NodeInfo.Tasks (map[string]*api.Task, nodeinfo.go:31): a node holds a few dozen tasks — a flat array searched front to back (one
// allocation, one cache line per probe) until it grows beyond TASKS_FLAT entries, a tree from then on.
class NodeTasks {
  public:
    static constexpr size_t TASKS_FLAT = 48;
    Value* find(const std::string& id) { return find(id, id_hash(id)); }
    Value* find(const std::string& id, uint64_t h) {
        if (!tree_.empty()) {
            auto it = tree_.find(id);
            return it == tree_.end() ? nullptr : &it->second;
        }
        // (the hashes lie together: a probe of a dozen tasks reads one cache line, and an entry only where the hash fits)
        for (size_t i = 0; i < hash_.size(); ++i)
            if (hash_[i] == (uint32_t)h && flat_[i].first == id) return &flat_[i].second;
        return nullptr;
    }
    void put(const std::string& id, const Value& t) { put(id, id_hash(id), t); }
    void put(const std::string& id, uint64_t h, const Value& t) {
        if (Value* have = find(id, h)) { *have = t; return; }
        if (tree_.empty() && flat_.size() < TASKS_FLAT) {
            if (flat_.capacity() == 0) { flat_.reserve(12); hash_.reserve(12); }   // (a node's first task: room for the usual dozen at once)
            flat_.emplace_back(id, t);
            hash_.push_back((uint32_t)h);
            return;
        }
        for (auto& kv : flat_) tree_.emplace(std::move(kv.first), std::move(kv.second));
        flat_.clear();
        hash_.clear();
        tree_.emplace(id, t);
    }
    bool erase(const std::string& id) {
        if (!tree_.empty()) return tree_.erase(id) != 0;
        const uint32_t h = (uint32_t)id_hash(id);
        for (size_t i = 0; i < flat_.size(); ++i)
            if (hash_[i] == h && flat_[i].first == id) {
                if (i + 1 != flat_.size()) { flat_[i] = std::move(flat_.back()); hash_[i] = hash_.back(); }
                flat_.pop_back();
                hash_.pop_back();
                return true;
            }
        return false;
    }
    void prefetch() const {   // what put() of a new task reads and writes
        __builtin_prefetch(hash_.data());
        __builtin_prefetch(flat_.data() + flat_.size(), 1);
    }
    // in id order (what ranging over a sorted key list gives: the outputs are deterministic)
    template <class F> void each_sorted(F f) const {
        if (!tree_.empty()) { for (const auto& kv : tree_) f(kv.first, kv.second); return; }
        std::vector<const std::pair<std::string, Value>*> p;
        for (const auto& kv : flat_) p.push_back(&kv);
        std::sort(p.begin(), p.end(), [](auto a, auto b) { return a->first < b->first; });
        for (auto q : p) f(q->first, q->second);
    }

  private:
    std::vector<std::pair<std::string, Value>> flat_;
    std::vector<uint32_t> hash_;   // id_hash of flat_[i].first, low half
    std::map<std::string, Value> tree_;
};

}   // namespace swp
