#include "gr_opts.hpp"

#include <cstdlib>
#include <getopt.h>
#include <iostream>

namespace gr {

void
print_usage(const std::string& progname)
{
  // the text (including its stale "[1]" / "[5]" defaults) is the reference's
  // user-visible interface, opt.cpp:36-87
  static const char* const lines[] = {
    "  -k K -w W -i INPUT -g G [-p prefix] [-P PHRED_AVG] [-o O] [-t T] [-f F] [-h H] [-u U] [-m M] [-H HASH_UNIVERSE] [-s S] [-x X] [-M MAX_PATHS][-a A] [-j J] [-b B] [-d D] [--silver_path] [--ntcard] [--help] \n\n",
    "  -i INPUT                find golden paths from INPUT [required]\n",
    "  -g G                    estimated genome size [required]\n",
    "  -b B                    during insertion, B number of consecutive tiles to be inserted with the same ID [10]\n",
    "  -d D                    remove reads with greater or equal then D phred average between first half and second half of the read [5]\n",
    "  -f F                    don't use reads from F. Expects one read per line\n",
    "  -o O                    use O as occupancy [0.1]\n",
    "  -h H                    use h as number of spaced seed patterns [1]\n",
    "  -H HASH_UNIVERSE        determine MiBF size based on HASH_UNIVERSE [Calculated based on W and h]\n",
    "  -t T                    tile length [1000]\n",
    "  -k K                    span of spaced seed [required]\n",
    "  -w W                    weight of spaced seed [required]\n",
    "  -m M                    use reads longer than M [20000]\n",
    "  -u U                    U minimum unassigned tiles for read to be unassigned [5]\n",
    "  -a A                    A maximum assigned tiles for read to be unassigned [1]\n",
    "  -p prefix               write output to files with prefix [goldrush_out]\n",
    "  -P PHRED_AVG            minimum average phred score for each read [0 (calculates phred score minimum automatically)]\n",
    "  -j J                    number of threads [48]\n",
    "  -s S                    use S seed preset. Must be consistent with k and w [n/a, generate one randomly based on k and w]\n",
    "  -x X                    require X hits for a tile to be assigned [10]\n",
    "  -M MAX_PATHS            output MAX_PATHS [5, used with --silver_path]\n",
    "  --ntcard                use ntcard to estimate genome size [false, assume max entries]\n",
    "  --silver_path           generate silver path(s) instead of golden path. Silver paths terminate when the number of bases recruited equals or exceeds T * r\n",
    " --verbose                print verbose messages [false]\n",
    "  --help                  display this help and exit\n",
  };
  std::cout << "Usage:  " << progname;
  for (const char* l : lines) {
    std::cout << l;
  }
  std::cout.flush();
}

int
process_options(Opts& o, int argc, char** argv)
{
  const struct option longopts[] = { { "debug", no_argument, &o.debug, 1 },       { "verbose", no_argument, &o.verbose, 1 },
                                     { "silver_path", no_argument, &o.silver_path, 1 }, { "help", no_argument, &o.help, 1 },
                                     { "ntcard", no_argument, &o.ntcard, 1 },     { nullptr, 0, nullptr, 0 } };
  optind = 0; // allow repeated parsing inside one process
  int c, idx = 0;
  while ((c = getopt_long(argc, argv, "a:b:d:f:g:h:i:j:k:m:M:o:r:s:t:u:w:x:p:P:H:", longopts, &idx)) != -1) {
    switch (c) {
      case 0: break;
      case 'a': o.assigned_max = strtoul(optarg, nullptr, 10); break;
      case 'b': o.block_size = strtoul(optarg, nullptr, 10); break;
      case 'd': o.phred_delta = (uint32_t)strtoul(optarg, nullptr, 10); break;
      case 'f': o.filter_file = optarg; break;
      case 'H': o.hash_universe = strtoull(optarg, nullptr, 10); break;
      case 'h': o.hash_num = strtoul(optarg, nullptr, 10); break;
      case 'i': o.input = optarg; break;
      case 'j': o.jobs = strtoul(optarg, nullptr, 10); break;
      case 'k': o.kmer_size = strtoul(optarg, nullptr, 10); break;
      case 'm': o.min_length = strtoul(optarg, nullptr, 10); break;
      case 'M': o.max_paths = strtoul(optarg, nullptr, 10); break;
      case 'o': o.occupancy = strtod(optarg, nullptr); break;
      case 'r': o.ratio = strtod(optarg, nullptr); break;
      case 'p': o.prefix_file = optarg; break;
      case 'P': o.phred_min = (uint32_t)strtoul(optarg, nullptr, 10); break;
      case 's': o.seed_preset = optarg; break;
      case 't': o.tile_length = strtoul(optarg, nullptr, 10); break;
      case 'g': o.genome_size = (uint64_t)strtod(optarg, nullptr); break;
      case 'u': o.unassigned_min = strtoul(optarg, nullptr, 10); break;
      case 'w': o.weight = strtoul(optarg, nullptr, 10); break;
      case 'x': o.threshold = strtoul(optarg, nullptr, 10); break;
      default: return EXIT_FAILURE;
    }
  }
  auto die = [](const char* msg) {
    std::cerr << msg << std::endl;
    print_usage("goldrush_path");
    return 1;
  };
  if (o.help) {
    print_usage("goldrush_path");
    return 0;
  }
  if (!o.kmer_size) {
    return die("span of spaced seed cannot be 0");
  }
  if (!o.weight) {
    return die("weight of spaced seed cannot be 0");
  }
  if (o.genome_size == 0) {
    return die("genome size cannot be 0");
  }
  if (!o.seed_preset.empty()) {
    if (o.kmer_size != o.seed_preset.size()) {
      return die("seed preset must be the same size of k");
    }
    uint8_t ones = 0; // uint8_t like the reference (wraps past 255)
    for (char ch : o.seed_preset) {
      if (ch == '1') {
        ++ones;
      }
    }
    if (o.weight != ones) {
      return die("seed preset must have the same weight as w");
    }
  }
  return -1;
}

} // namespace gr
