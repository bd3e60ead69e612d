// CPU check of the facade's communicator backends (Teuchos_shim.hpp): the same collectives over threads of one process
// (runAsRanks) and over processes that meet through files (RANK / WORLD_SIZE / FEDD_RENDEZVOUS).  Prints one line per rank.
#include <cstring>
#include <iostream>

#include "Teuchos_shim.hpp"

static int body() {
    auto comm = Teuchos::DefaultComm<int>::getComm();
    const int rank = comm->getRank(), size = comm->getSize();
    unsigned char id[128];
    for (int i = 0; i < 128; ++i) id[i] = rank == 0 ? (unsigned char)(7 * i + 3) : 0;
    comm->broadcast(0, sizeof(id), id);
    int bad = 0;
    for (int i = 0; i < 128; ++i) bad += id[i] != (unsigned char)(7 * i + 3);
    double v[3] = {1.0, (double)rank, 0.5 * rank * rank};
    for (int rep = 0; rep < 50; ++rep) {       // many rounds: the file backend recycles its files
        double w[3] = {v[0], v[1], v[2]};
        comm->sumAll(w, 3);
        if (w[0] != size || w[1] != 0.5 * size * (size - 1)) ++bad;
    }
    comm->barrier();
    std::vector<char> all;
    long mine = 100 + rank;
    comm->gatherAll(&mine, sizeof(mine), all);
    for (int r = 0; r < size; ++r) bad += ((const long*)all.data())[r] != 100 + r;
    if (size > 1 && comm->backend()->inProcess()) {     // point-to-point ring of the thread backend
        double out = 10.0 + rank, in = -1.0;
        comm->backend()->send(rank, (rank + 1) % size, &out, 1);
        comm->backend()->recv((rank + size - 1) % size, rank, &in, 1);
        bad += in != 10.0 + (rank + size - 1) % size;
    }
    {   // one write per rank: the ranks may be threads of this process
        static std::mutex out;
        std::lock_guard<std::mutex> lk(out);
        std::cout << "rank " << rank << " of " << size << " bad " << bad << std::endl;
    }
    return bad;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strncmp(argv[1], "--threads=", 10) == 0) return Teuchos::runAsRanks(std::atoi(argv[1] + 10), [](int) { return body(); });
    return body();
}
