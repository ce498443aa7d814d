// tools/host_layer_prof.cpp — the JSON host layer (csrc/swp_sched.cpp) over the engine double (tests/fake_swp.cpp) as ONE native
// executable, for profilers that want a plain process (gprof, a sampling debugger): the same events tools/host_layer_bench.py sends,
// read from files. Development tool; not part of the product or of any test.
//   python tools/host_layer_bench.py --dump /tmp/hl        (nodes.jsonl, services.txt, tasks.jsonl)
//   g++ -std=c++17 -O2 -g -pg -Iinclude -o /tmp/hl/prof tools/host_layer_prof.cpp tests/fake_swp.cpp swarmkit_amd/csrc/swp_sched.cpp
//   /tmp/hl/prof /tmp/hl [repetitions] [all|nodes|events|tick] && gprof /tmp/hl/prof gmon.out | head -40
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

extern "C" int moncontrol(int);   // glibc: gprof sampling on / off

#include "swp.h"
#include "swp_sched.h"

static std::vector<std::string> lines(const std::string& path) {
    std::ifstream f(path);
    std::vector<std::string> out;
    for (std::string l; std::getline(f, l);)
        if (!l.empty()) out.push_back(l);
    return out;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/hl";
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1;
    const std::string phase = argc > 3 ? argv[3] : "all";   // what gprof samples: all | nodes | events | tick
    moncontrol(phase == "all");
    const auto nodes = lines(dir + "/nodes.jsonl"), services = lines(dir + "/services.txt"), tasks = lines(dir + "/tasks.jsonl");
    for (int r = 0; r < reps; ++r) {
        swp_engine* e = nullptr;
        swp_config cfg{};
        if (swp_create(&cfg, &e) != SWP_OK) { std::fprintf(stderr, "swp_create: %s\n", swp_last_error(nullptr)); return 1; }
        swp_sched* s = nullptr;
        if (swp_sched_create(e, &s) != SWP_OK) { std::fprintf(stderr, "swp_sched_create: %s\n", swp_sched_last_error(nullptr)); return 1; }
        moncontrol(phase == "all" || phase == "nodes");
        double t0 = now();
        for (const auto& n : nodes)
            if (swp_sched_create_or_update_node(s, n.data(), n.size()) != SWP_OK) { std::fprintf(stderr, "node: %s\n", swp_sched_last_error(s)); return 1; }
        for (const auto& v : services) swp_sched_set_service(s, v.data(), v.size(), 0, 0);
        double t1 = now();
        int flag = 0;
        moncontrol(phase == "all" || phase == "events");
        for (const auto& t : tasks)
            if (swp_sched_create_task(s, t.data(), t.size(), &flag) != SWP_OK) { std::fprintf(stderr, "task: %s\n", swp_sched_last_error(s)); return 1; }
        double t2 = now();
        moncontrol(phase == "all" || phase == "tick");
        const char* out = nullptr;
        int rc = swp_sched_tick(s, &out);
        double t3 = now();
        moncontrol(phase == "all");
        std::printf("nodes %.3f s, create_task %.3f s (%.2f us each), tick %.3f s (%.2f us per task) rc %d, %zu bytes\n", t1 - t0, t2 - t1, (t2 - t1) / tasks.size() * 1e6, t3 - t2,
                    (t3 - t2) / tasks.size() * 1e6, rc, out ? std::char_traits<char>::length(out) : 0);
        swp_sched_destroy(s);
        swp_destroy(e);
    }
    return 0;
}
