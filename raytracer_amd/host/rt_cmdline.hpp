// rt_cmdline.hpp -- the slice of cli::Parser that raytracer_cpp/main.cpp:232-238 uses:
// set_optional<T>(short, long, default, description), run(), get<T>(short).  The reference takes
// it from the CmdParser submodule, which is not vendored (.gitmodules:1-3), so spellings beyond
// "-p 2" / "--partitions 2" / "-p=2" and its help/error text are parity-unpinned.
#pragma once
#include <cstdlib>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace cli {

class Parser {
public:
    Parser(int argc, char **argv)
    {
        for (int i = 0; i < argc; i++) args_.push_back(argv[i]);
    }

    template <typename T>
    void set_optional(const std::string &shortName, const std::string &longName, T defaultValue,
                      const std::string &description = "")
    {
        std::ostringstream os;
        os << defaultValue;
        options_.push_back({shortName, longName, description, os.str()});
        values_[shortName] = os.str();
    }

    // false (after printing usage) on -h/--help or a malformed option
    bool run(std::ostream &err = std::cerr)
    {
        for (size_t i = 1; i < args_.size(); i++) {
            std::string a = args_[i], value;
            bool inlineValue = false;
            const size_t eq = a.find('=');
            if (eq != std::string::npos) {
                value = a.substr(eq + 1);
                a = a.substr(0, eq);
                inlineValue = true;
            }
            if (a == "-h" || a == "--help") return usage(err), false;
            const Option *opt = nullptr;
            for (const Option &o : options_)
                if (a == "-" + o.shortName || a == "--" + o.longName) opt = &o;
            if (!opt) {
                err << "unknown option '" << args_[i] << "'\n";
                return usage(err), false;
            }
            if (!inlineValue) {
                if (i + 1 >= args_.size()) {
                    err << "option '" << a << "' needs a value\n";
                    return usage(err), false;
                }
                value = args_[++i];
            }
            values_[opt->shortName] = value;
        }
        return true;
    }

    template <typename T>
    T get(const std::string &shortName) const
    {
        T v{};
        auto it = values_.find(shortName);
        if (it != values_.end()) {
            std::istringstream is(it->second);
            is >> v;
        }
        return v;
    }

private:
    struct Option {
        std::string shortName, longName, description, defaultText;
    };
    void usage(std::ostream &err) const
    {
        err << "usage: " << (args_.empty() ? "raytracer" : args_[0]) << " [options]\n";
        for (const Option &o : options_)
            err << "  -" << o.shortName << ", --" << o.longName << "  " << o.description << " (default " << o.defaultText << ")\n";
    }
    std::vector<std::string> args_;
    std::vector<Option> options_;
    std::map<std::string, std::string> values_;
};

} // namespace cli
