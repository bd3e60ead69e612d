// Laplace driver on the MI355X path, written against the FEDD:: operator surface exactly the way
// the reference's driver is (feddlib/problems/tests/laplace/main.cpp:56-228): same XML parameter
// files, same call sequence  Domain::buildMesh -> BCBuilder::addBC -> Laplace(...) ->
// addRhsFunction -> addBoundaries -> initializeProblem -> assemble -> setBoundaries -> solve.
// Output: iteration count on stdout, solution as text (--out, for the tests) and, like the reference's tail
// (main.cpp:210-225), through ExporterParaView: solutionLaplace.xmf + raw binary data in the working directory.
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>

#include "feddlib/core/FEDDCore.hpp"
#include "feddlib/core/FE/Domain.hpp"
#include "feddlib/core/General/BCBuilder.hpp"
#include "feddlib/core/General/ExporterParaView.hpp"
#include "feddlib/core/Mesh/MeshPartitioner.hpp"
#include "feddlib/problems/specific/Laplace.hpp"

void zeroBC(double* x, double* res, double t, const double* parameters) { res[0] = 0.; }
void zeroBC2D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; }
void zeroBC3D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; res[2] = 0.; }
void oneFunc(double* x, double* res, double* parameters) { res[0] = 1.; }

typedef default_sc SC;
typedef default_lo LO;
typedef default_go GO;
typedef default_no NO;

using namespace FEDD;

// the body every rank runs (the reference's main between MPI_Init and MPI_Finalize)
static int run(int argc, char* argv[]) {
    std::string xmlProblemFile = "parametersProblem.xml", xmlPrecFile = "parametersPrec.xml", xmlSolverFile = "parametersSolver.xml";
    std::string outFile = "solutionLaplace.txt";
    double length = 4.;
    bool vL = false, userPrec = false;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        auto val = [&](const char* key, std::string& dst) {
            const std::string k = std::string("--") + key + "=";
            if (a.compare(0, k.size(), k) == 0) { dst = a.substr(k.size()); return true; }
            return false;
        };
        std::string tmp;
        if (val("problemfile", xmlProblemFile) || val("precfile", xmlPrecFile) || val("solverfile", xmlSolverFile) || val("out", outFile)) continue;
        if (val("length", tmp)) { length = std::atof(tmp.c_str()); continue; }
        if (a == "--vectorLaplace") { vL = true; continue; }
        if (val("ranks-as-threads", tmp)) continue;     // handled by main
        if (a == "--user-preconditioner") { userPrec = true; continue; }
        std::cerr << "unknown option " << a << std::endl;
        return 2;
    }
    (void)length;
    try {
        Teuchos::RCP<const Teuchos::Comm<int> > comm = Teuchos::DefaultComm<int>::getComm();
        ParameterListPtr_Type parameterListProblem = Teuchos::getParametersFromXmlFile(xmlProblemFile);
        ParameterListPtr_Type parameterListPrec = Teuchos::getParametersFromXmlFile(xmlPrecFile);
        ParameterListPtr_Type parameterListSolver = Teuchos::getParametersFromXmlFile(xmlSolverFile);
        ParameterListPtr_Type parameterListAll(new Teuchos::ParameterList(*parameterListProblem));
        parameterListAll->setParameters(*parameterListPrec);
        parameterListAll->setParameters(*parameterListSolver);

        int dim = parameterListProblem->sublist("Parameter").get("Dimension", 2);
        int m = parameterListProblem->sublist("Parameter").get("H/h", 5);
        std::string FEType = parameterListProblem->sublist("Parameter").get("Discretization", "P1");
        std::string meshType = parameterListProblem->sublist("Parameter").get("Mesh Type", "structured");
        int numProcsCoarseSolve = parameterListProblem->sublist("General").get("Mpi Ranks Coarse", 0);
        int size = comm->getSize() - numProcsCoarseSolve;

        Teuchos::RCP<Domain<SC, LO, GO, NO> > domain;
        TEUCHOS_TEST_FOR_EXCEPTION(meshType != "structured" && meshType != "unstructured", std::logic_error,
                                   "this driver builds 'structured' meshes and reads 'unstructured' ones");
        int n;
        if (meshType == "unstructured") {       // laplace/main.cpp:155-175: read, partition over the ranks, P1
            Teuchos::RCP<Domain<SC, LO, GO, NO> > domainP1(new Domain<SC, LO, GO, NO>(comm, dim));
            MeshPartitioner<SC, LO, GO, NO>::DomainPtrArray_Type domainP1Array(1);
            domainP1Array[0] = domainP1;
            ParameterListPtr_Type pListPartitioner(new Teuchos::ParameterList(parameterListAll->sublist("Mesh Partitioner")));
            MeshPartitioner<SC, LO, GO, NO> partitionerP1(domainP1Array, pListPartitioner, "P1", dim);
            partitionerP1.readAndPartition();
            TEUCHOS_TEST_FOR_EXCEPTION(FEType != "P1", std::logic_error, "unstructured meshes: P1 in this driver");
            domain = domainP1;
        } else if (dim == 2) {
            n = (int)(std::pow(size, 1 / 2.) + 100. * 2.220446049250313e-16);
            std::vector<double> x(2);
            x[0] = 0.0; x[1] = 0.0;
            domain = Teuchos::rcp(new Domain<SC, LO, GO, NO>(x, 1., 1., comm));
            domain->buildMesh(1, "Square", dim, FEType, n, m, numProcsCoarseSolve);
        } else {
            n = (int)(std::pow(size, 1 / 3.) + 100. * 2.220446049250313e-16);
            std::vector<double> x(3);
            x[0] = 0.0; x[1] = 0.0; x[2] = 0.0;
            domain = Teuchos::rcp(new Domain<SC, LO, GO, NO>(x, 1., 1., 1., comm));
            domain->buildMesh(1, "Square", dim, FEType, n, m, numProcsCoarseSolve);
        }

        Teuchos::RCP<BCBuilder<SC, LO, GO, NO> > bcFactory(new BCBuilder<SC, LO, GO, NO>());
        if (vL) {
            BC_func_Type z = dim == 2 ? zeroBC2D : zeroBC3D;
            bcFactory->addBC(z, 1, 0, domain, "Dirichlet", dim);
            bcFactory->addBC(z, 2, 0, domain, "Dirichlet", dim);
            bcFactory->addBC(z, 3, 0, domain, "Dirichlet", dim);
        } else {
            bcFactory->addBC(zeroBC, 1, 0, domain, "Dirichlet", 1);
            bcFactory->addBC(zeroBC, 2, 0, domain, "Dirichlet", 1);
            bcFactory->addBC(zeroBC, 3, 0, domain, "Dirichlet", 1);
        }

        Laplace<SC, LO, GO, NO> laplace(domain, FEType, parameterListAll, vL);
        int its;
        {
            laplace.addRhsFunction(oneFunc);
            laplace.addBoundaries(bcFactory);

            laplace.initializeProblem();
            laplace.assemble();
            laplace.setBoundaries();
            if (userPrec) {
                // the iterative solver stays on the host and gets the GPU's Schwarz preconditioner as
                // a PreconditionerOperator (Problem::setPreconditionerThyraFromLinOp, Problem_def.hpp:397-399)
                auto& fr = parameterListAll->sublist("ThyraPreconditioner").sublist("Preconditioner Types").sublist("FROSch");
                std::string combine = fr.sublist("AlgebraicOverlappingOperator").get("Combine Values in Overlap", "Restricted");
                Teuchos::RCP<PreconditionerOperator<SC, LO, GO, NO> > op(
                    new DeviceSchwarzOperator<SC, LO, GO, NO>(domain->device(), fr.get("Overlap", 1), combine, fr.get("TwoLevel", false)));
                laplace.setPreconditionerThyraFromLinOp(op);
            }
            its = laplace.solve();
        }
        if (comm->getRank() == 0) std::cout << "iterations " << its << " relres " << laplace.getLastRelativeResidual() << std::endl;

        Teuchos::RCP<const MultiVector<SC, LO, GO, NO> > exportSolution = laplace.getSolution()->getBlock(0);
        // several ranks: every rank writes the entries of its unique map (<out>.<rank>)
        std::ofstream out(comm->getSize() > 1 ? outFile + "." + std::to_string(comm->getRank()) : outFile);
        out << std::setprecision(17);
        auto map = exportSolution->getMap();
        auto data = exportSolution->getData(0);
        for (size_t i = 0; i < data.size(); ++i) out << map->getGlobalElement((LO)i) << " " << data[i] << "\n";

        bool boolExportSolution = true;     // (laplace/main.cpp:210-225; every rank writes its part of the global arrays)
        if (boolExportSolution) {
            Teuchos::RCP<ExporterParaView<SC, LO, GO, NO> > exPara(new ExporterParaView<SC, LO, GO, NO>());
            exPara->setup("solutionLaplace", domain->getMesh(), FEType);
            if (vL)
                exPara->addVariable(exportSolution, "u", "Vector", dim, domain->getMapUnique());
            else
                exPara->addVariable(exportSolution, "u", "Scalar", 1, domain->getMapUnique());
            exPara->save(0.0);
        }
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
