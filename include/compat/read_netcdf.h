/* include/compat/read_netcdf.h -- forwarding header: lets host code written against the reference's
 * header names (src/read_netcdf.h) compile unchanged against the drop-in boundary.
 * Use: cc -Iinclude/compat -Iinclude ... -lcfdproxy_hip */
#include "../cfdproxy_dropin.h"
