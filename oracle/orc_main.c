/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  CLI wrapper around orc_main():
 * same flags and output files as goldrush-path (goldrush_path.cpp:1096-1275).
 * PARITY UNPINNED (see orc_path.h).
 */
#include "orc_path.h"

int
main(int argc, char** argv)
{
  return orc_main(argc, argv);
}
