# SearchFactHIPFACT.cmake — locates the hipfact library for -DSLEQP_FACT=HIPFACT.
#
# Copy to <sleqp>/cmake/ and register the backend in cmake/SearchFact.cmake with
#   add_fact(NAME "HIPFACT" SOURCES fact/fact_hipfact.c)
# (mechanism: cmake/SearchFact.cmake:11-83; required variables :105-117).
# The C shim is compiled by the host C compiler; the HIP code lives in
# libhipfact.so, named by HIPFACT_LIBRARIES.

find_path(HIPFACT_INCLUDE_DIRS
  NAMES hipfact.h
  HINTS ${HIPFACT_ROOT} $ENV{HIPFACT_ROOT}
  PATH_SUFFIXES include)

find_library(HIPFACT_LIBRARIES
  NAMES hipfact
  HINTS ${HIPFACT_ROOT} $ENV{HIPFACT_ROOT}
  PATH_SUFFIXES lib sleqp_amd/csrc)

if(HIPFACT_INCLUDE_DIRS AND HIPFACT_LIBRARIES)
  file(STRINGS "${HIPFACT_INCLUDE_DIRS}/hipfact.h" _hipfact_version_line
    REGEX "#define HIPFACT_VERSION ")
  string(REGEX REPLACE ".*\"(.*)\".*" "\\1" HIPFACT_VERSION "${_hipfact_version_line}")
  get_filename_component(HIPFACT_LIBRARY_DIRS "${HIPFACT_LIBRARIES}" DIRECTORY)
  set(HIPFACT_DEFINITIONS "")
endif()

include(FindPackageHandleStandardArgs)
find_package_handle_standard_args(HIPFACT
  REQUIRED_VARS HIPFACT_LIBRARIES HIPFACT_INCLUDE_DIRS
  VERSION_VAR HIPFACT_VERSION)
