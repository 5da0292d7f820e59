// tclap_ref.cpp — the reference's own command-line library at work.  TEST INFRASTRUCTURE ONLY.
//
// The reference's tools parse their arguments with the header-only TCLAP that is vendored under include/tclap
// (tools/dosplitalign.cpp:43-71 and the other mains).  This driver builds a TCLAP::CmdLine from a specification given on
// its own command line and lets TCLAP parse the remaining arguments — so usage texts, PARSE ERROR messages and exit codes are
// TCLAP's, and tests/test_cli_ref.py can hold the tools' own parser (tools_src/defuse_host.hpp:CmdLine) against them.
// Compiled by oracle/Makefile against the headers where they lie (-I/root/reference/include) into oracle/_ref/tclap_ref;
// nothing of TCLAP is copied into this repository.
//
//   tclap_ref <program name> <message> <n> { <flag> <name> <description> <type> <required 0|1> } x n -- <arguments to parse>
//   <type> = string|int|float|switch, optionally followed by ":<label>", the type description TCLAP shows in its usage text
//   (the reference labels some double arguments "integer", tools/clustermatepairs.cpp:403-404)
//
// On success it prints one line per argument: name, tab, "1"/"0" for a switch or the parsed value.
#include <cstdio>              // EOF for the vendored headers (the reference's tools get it through <iostream> of their day)
#include <tclap/CmdLine.h>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

int main(int argc, char* argv[])
{
    if (argc < 4) return 2;
    const std::string prog = argv[1], message = argv[2];
    const int n = std::atoi(argv[3]);
    int at = 4;
    struct Spec { std::string flag, name, desc, type, label; bool req; };
    std::vector<Spec> specs;
    for (int k = 0; k < n; ++k) {
        if (at + 5 > argc) return 2;
        std::string type = argv[at + 3], label;
        const size_t colon = type.find(':');
        if (colon != std::string::npos) { label = type.substr(colon + 1); type = type.substr(0, colon); }
        if (label.empty()) label = type == "int" ? "integer" : type;
        specs.push_back(Spec{argv[at], argv[at + 1], argv[at + 2], type, label, std::atoi(argv[at + 4]) != 0});
        at += 5;
    }
    if (at >= argc || std::string(argv[at]) != "--") return 2;
    ++at;
    std::vector<const char*> rest;
    rest.push_back(prog.c_str());
    for (; at < argc; ++at) rest.push_back(argv[at]);
    try {
        TCLAP::CmdLine cmd(message);
        std::vector<std::unique_ptr<TCLAP::Arg>> args;
        for (const Spec& s : specs) {
            if (s.type == "switch") args.emplace_back(new TCLAP::SwitchArg(s.flag, s.name, s.desc, cmd));
            else if (s.type == "int") args.emplace_back(new TCLAP::ValueArg<int>(s.flag, s.name, s.desc, s.req, 0, s.label, cmd));
            else if (s.type == "float") args.emplace_back(new TCLAP::ValueArg<double>(s.flag, s.name, s.desc, s.req, 0.0, s.label, cmd));
            else args.emplace_back(new TCLAP::ValueArg<std::string>(s.flag, s.name, s.desc, s.req, "", s.label, cmd));
        }
        cmd.parse((int)rest.size(), const_cast<char**>(rest.data()));
        for (size_t k = 0; k < specs.size(); ++k) {
            std::cout << specs[k].name << "\t";
            if (specs[k].type == "switch") std::cout << (static_cast<TCLAP::SwitchArg*>(args[k].get())->getValue() ? 1 : 0);
            else if (specs[k].type == "int") std::cout << static_cast<TCLAP::ValueArg<int>*>(args[k].get())->getValue();
            else if (specs[k].type == "float") std::cout << static_cast<TCLAP::ValueArg<double>*>(args[k].get())->getValue();
            else std::cout << static_cast<TCLAP::ValueArg<std::string>*>(args[k].get())->getValue();
            std::cout << "\n";
        }
    } catch (TCLAP::ArgException& e) {             // the handler every tool of the reference has (e.g. tools/dosplitalign.cpp:73-77)
        std::cerr << "Error: " << e.error() << " for arg " << e.argId() << std::endl;
        return 1;
    }
    return 0;
}
