// json_echo.cpp — parse every line of stdin with the host layer's JSON reader (swarmkit_amd/csrc/swp_json.hpp) and write it back with its
// writer: tests/test_host_json_roundtrip_cpu.py compares what comes back with what Python's json module makes of the same text.
// A line the reader refuses comes back as "!<reason>". TEST INFRASTRUCTURE.
#include <iostream>
#include <string>

#include "../../swarmkit_amd/csrc/swp_json.hpp"

int main() {
    std::string line, out;
    while (std::getline(std::cin, line)) {
        try {
            const swp::json::Value v = swp::json::parse(line);
            out.clear();
            swp::json::dump(out, v);
            std::cout << out << "\n";
        } catch (const swp::json::ParseError& e) {
            std::cout << "!" << e.what() << "\n";
        }
    }
    return 0;
}
