// InputParser.h — command-line token lookup with the reference's two-call surface
// (RestOfLife/InputParser.h:9-30: getCmdOption / cmdOptionExists), written from scratch.
#pragma once
#include <string>
#include <vector>

class InputParser {
public:
    InputParser(int& argc, char** argv) {
        tokens_.reserve(argc > 1 ? static_cast<size_t>(argc - 1) : 0u);
        for (int i = 1; i < argc; ++i) tokens_.emplace_back(argv[i]);
    }

    // value following the first occurrence of `option`, or "" when the option is absent or last
    const std::string& getCmdOption(const std::string& option) const {
        static const std::string none;
        for (size_t i = 0; i + 1 < tokens_.size(); ++i)
            if (tokens_[i] == option) return tokens_[i + 1];
        return none;
    }

    bool cmdOptionExists(const std::string& option) const {
        for (const std::string& t : tokens_)
            if (t == option) return true;
        return false;
    }

private:
    std::vector<std::string> tokens_;
};
