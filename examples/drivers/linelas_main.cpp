// Steady linear elasticity driver on the MI355X path, written against the FEDD:: operator surface
// the way the reference's performance driver is
// (feddlib/problems/tests/steadyLinElas_Perf/main.cpp:69-252): same XML parameter files
// (parametersProblem / parametersPrec / parametersSolver), same call sequence
//   Domain::buildMesh -> BCBuilder::addBC(zero Dirichlet on "Homogeneous Dirichlet Flag") ->
//   LinElas(...) -> addRhsFunction -> addBoundaries -> addParemeterRhs(force, degree) ->
//   initializeProblem -> assemble -> setBoundaries -> solve.
// Output: iteration count and relative residual on stdout, the displacement as text (the reference
// writes HDF5/XDMF through ExporterParaView and prints a Teuchos::StackedTimer report, both out of
// scope here; a wall-clock time of the assemble + solve section is printed instead).
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>

#include "feddlib/core/FEDDCore.hpp"
#include "feddlib/core/FE/Domain.hpp"
#include "feddlib/core/General/BCBuilder.hpp"
#include "feddlib/core/General/ExporterParaView.hpp"
#include "feddlib/problems/specific/LinElas.hpp"

void zeroDirichlet2D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; }
void zeroDirichlet3D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; res[2] = 0.; }
// parameters[0] is the time, parameters[1] the volume force
void rhs2D(double* x, double* res, double* parameters) { res[0] = 0.; res[1] = parameters[1]; }
void rhs3D(double* x, double* res, double* parameters) { res[0] = 0.; res[1] = parameters[1]; res[2] = 0.; }

typedef default_sc SC;
typedef default_lo LO;
typedef default_go GO;
typedef default_no NO;

using namespace FEDD;

// the body every rank runs (the reference's main between MPI_Init and MPI_Finalize)
static int run(int argc, char* argv[]) {
    std::string xmlProblemFile = "parametersProblem.xml", xmlPrecFile = "parametersPrec.xml", xmlSolverFile = "parametersSolver.xml";
    std::string outFile = "solutionLinElas.txt";
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        auto val = [&](const char* key, std::string& dst) {
            const std::string k = std::string("--") + key + "=";
            if (a.compare(0, k.size(), k) == 0) { dst = a.substr(k.size()); return true; }
            return false;
        };
        if (val("problemfile", xmlProblemFile) || val("precfile", xmlPrecFile) || val("solverfile", xmlSolverFile) || val("out", outFile)) continue;
        std::string tmp;
        if (val("ranks-as-threads", tmp)) continue;     // handled by main
        std::cerr << "unknown option " << a << std::endl;
        return 2;
    }
    try {
        Teuchos::RCP<const Teuchos::Comm<int> > comm = Teuchos::DefaultComm<int>::getComm();
        const bool root = comm->getRank() == 0;
        // steadyLinElas_Perf/main.cpp:114-115: every FEDD timer goes into one stacked timer, reported at the end
        Teuchos::RCP<Teuchos::StackedTimer> stackedTimer = Teuchos::rcp(new Teuchos::StackedTimer("Steady Linear Elasticity Performance Test"));
        Teuchos::TimeMonitor::setStackedTimer(stackedTimer);
        ParameterListPtr_Type parameterListProblem = Teuchos::getParametersFromXmlFile(xmlProblemFile);
        ParameterListPtr_Type parameterListPrec = Teuchos::getParametersFromXmlFile(xmlPrecFile);
        ParameterListPtr_Type parameterListSolver = Teuchos::getParametersFromXmlFile(xmlSolverFile);
        ParameterListPtr_Type parameterListAll(new Teuchos::ParameterList(*parameterListProblem));
        parameterListAll->setParameters(*parameterListPrec);
        parameterListAll->setParameters(*parameterListSolver);

        int dim = parameterListProblem->sublist("Parameter").get("Dimension", 2);
        int m = parameterListProblem->sublist("Parameter").get("H/h", 5);
        int zeroDirID = parameterListProblem->sublist("Parameter").get("Homogeneous Dirichlet Flag", 1);
        std::string discType = parameterListProblem->sublist("Parameter").get("Discretization", "P2");
        int numProcsCoarseSolve = parameterListProblem->sublist("General").get("Mpi Ranks Coarse", 0);
        int size = comm->getSize() - numProcsCoarseSolve;

        Teuchos::RCP<Domain<SC, LO, GO, NO> > domain;
        int n;
        if (dim == 2) {
            n = (int)(std::pow(size, 1 / 2.) + 100. * 2.220446049250313e-16);
            std::vector<double> x(2);
            x[0] = 0.0; x[1] = 0.0;
            domain = Teuchos::rcp(new Domain<SC, LO, GO, NO>(x, 1., 1., comm));
            domain->buildMesh(1, "Square", dim, discType, n, m, 0);
        } else {
            n = (int)(std::pow(size, 1 / 3.) + 100. * 2.220446049250313e-16);
            std::vector<double> x(3);
            x[0] = 0.0; x[1] = 0.0; x[2] = 0.0;
            domain = Teuchos::rcp(new Domain<SC, LO, GO, NO>(x, 1., 1., 1., comm));
            domain->buildMesh(1, "Square", dim, discType, n, m, numProcsCoarseSolve);
        }

        Teuchos::RCP<BCBuilder<SC, LO, GO, NO> > bcFactory(new BCBuilder<SC, LO, GO, NO>());
        bcFactory->addBC(dim == 2 ? zeroDirichlet2D : zeroDirichlet3D, zeroDirID, 0, domain, "Dirichlet", dim);

        LinElas<SC, LO, GO, NO> linElas(domain, discType, parameterListAll);
        linElas.addRhsFunction(dim == 2 ? rhs2D : rhs3D);
        int its;
        const auto t0 = std::chrono::steady_clock::now();
        {
            linElas.addBoundaries(bcFactory);
            const double force = parameterListAll->sublist("Parameter").get("Volume force", 0.);
            const double degree = 0;
            linElas.addParemeterRhs(force);
            linElas.addParemeterRhs(degree);

            fedd_timing_enable(domain->device()->ctx, 1);
            linElas.initializeProblem();
            {
                Teuchos::TimeMonitor tm(*Teuchos::TimeMonitor::getNewCounter("Assemble Problem"));
                linElas.assemble();
                linElas.setBoundaries();
            }
            {
                Teuchos::TimeMonitor tm(*Teuchos::TimeMonitor::getNewCounter("FEDD - Problem - Solve"));
                its = linElas.solve();
                addDeviceTimers(domain->device(), *stackedTimer);      // where the GPU time of assemble + solve went
            }
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (root) {
            std::cout << "iterations " << its << " relres " << linElas.getLastRelativeResidual() << std::endl;
            std::cout << "Solve Problem " << secs << " s" << std::endl;
        }

        Teuchos::RCP<const MultiVector<SC, LO, GO, NO> > exportSolution = linElas.getSolution()->getBlock(0);
        // several ranks: every rank writes the entries of its unique map (<out>.<rank>)
        std::ofstream out(comm->getSize() > 1 ? outFile + "." + std::to_string(comm->getRank()) : outFile);
        out << std::setprecision(17);
        auto map = exportSolution->getMap();
        auto data = exportSolution->getData(0);
        for (size_t i = 0; i < data.size(); ++i) out << map->getGlobalElement((LO)i) << " " << data[i] << "\n";

        if (parameterListAll->sublist("General").get("ParaViewExport", false)) {       // main.cpp:228-240 (every rank writes its part)
            Teuchos::RCP<ExporterParaView<SC, LO, GO, NO> > exPara(new ExporterParaView<SC, LO, GO, NO>());
            exPara->setup("displacements", domain->getMesh(), discType);
            exPara->addVariable(exportSolution, "values", "Vector", dim, domain->getMapUnique());
            exPara->save(0.0);
            exPara->closeExporter();
        }
        comm->barrier();
        stackedTimer->stop("Steady Linear Elasticity Performance Test");                 // main.cpp:245-249
        Teuchos::StackedTimer::OutputOptions options;
        options.output_fraction = options.output_histogram = options.output_minmax = true;
        if (root) stackedTimer->report(std::cout, comm, options);
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

// ranks = processes under a launcher (RANK / WORLD_SIZE and FEDD_RENDEZVOUS in the environment, RCCL between the GPUs),
// or, with --ranks-as-threads=N, N threads of this process sharing one GPU (functional runs of the N > 1 path)
int main(int argc, char* argv[]) {
    Teuchos::GlobalMPISession mpiSession(&argc, &argv);
    int threads = 0;
    for (int i = 1; i < argc; ++i)
        if (std::strncmp(argv[i], "--ranks-as-threads=", 19) == 0) threads = std::atoi(argv[i] + 19);
    if (threads > 1) {
        try {
            return Teuchos::runAsRanks(threads, [&](int) { return run(argc, argv); });
        } catch (const std::exception& e) {
            std::cerr << "exception: " << e.what() << std::endl;
            return 1;
        }
    }
    return run(argc, argv);
}
