// Logging macros of the rebvio API (the reference wraps spdlog, util/log.hpp:58-69); here: plain stderr, no dependency.
#pragma once

#include <cstdio>

#define REBVIO_INFO(...)                  \
  do {                                    \
    std::fprintf(stderr, "[Rebvio] [info]: "); \
    std::fprintf(stderr, __VA_ARGS__);    \
    std::fprintf(stderr, "\n");           \
  } while (0)
#define REBVIO_WARN(...) REBVIO_INFO(__VA_ARGS__)
#define REBVIO_ERROR(...) REBVIO_INFO(__VA_ARGS__)
