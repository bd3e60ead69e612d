// forwarding header: the reference include path of ExporterParaView (feddlib/core/General/ExporterParaView.hpp)
#include "../../fedd_facade.hpp"
