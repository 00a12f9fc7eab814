/* include/compat/exchange_data_mpidma.h -- forwarding header (reference src/exchange_data_mpidma.h): the exchange back-ends the reference's
 * harness includes are replaced by the xGMI data path inside the library; nothing of them is called
 * from host code.  Use: cc -Iinclude/compat -Iinclude ... -lcfdproxy_mpi -lcfdproxy_hip */
#include "../cfdproxy_dropin.h"
